"""The communicator-id rendezvous of the C++ multi-GPU host (pcl_tracking_amd/include/pft/id_exchange.hpp,
examples/dist_tracking_amd.cpp): a record left by a crashed run, by another launch, or garbage at the path is never
taken for this launch's id.  CPU only: the header has no RCCL / HIP types."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("idx") / "id_exchange_tool")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "pcl_tracking_amd", "include"),
                    os.path.join(ROOT, "tests", "cpp", "id_exchange_tool.cpp"), "-o", out], check=True)
    return out


def _read(tool, path, nonce, timeout_ms=100):
    r = subprocess.run([tool, "read", path, str(nonce), str(timeout_ms)], capture_output=True, text=True)
    return r.returncode, r.stdout.strip()


def test_live_publisher_is_accepted_and_a_dead_one_is_not(tool, tmp_path):
    path = str(tmp_path / "id")
    pub = subprocess.Popen([tool, "publish", path, "42", "30000"], stdout=subprocess.PIPE, text=True)
    try:
        assert pub.stdout.readline().strip() == "published"
        assert _read(tool, path, 42) == (0, "ok 0 1 127")
        assert _read(tool, path, 43) == (3, "a record of another launch (nonce differs)")
    finally:
        pub.kill()
        pub.wait()
    # the file is still there, its publisher is not: what a crashed rank 0 leaves behind
    assert os.path.exists(path)
    rc, msg = _read(tool, path, 42)
    assert rc == 3 and "stale" in msg


def test_reader_that_starts_first_waits_for_the_record(tool, tmp_path):
    path = str(tmp_path / "id")
    rd = subprocess.Popen([tool, "read", path, "7", "5000"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.2)
    pub = subprocess.Popen([tool, "publish", path, "7", "3000"], stdout=subprocess.PIPE, text=True)
    try:
        out, _ = rd.communicate(timeout=10)
        assert rd.returncode == 0 and out.strip() == "ok 0 1 127"
    finally:
        pub.kill()
        pub.wait()


def test_stale_record_is_replaced_by_the_new_publisher(tool, tmp_path):
    path = str(tmp_path / "id")
    old = subprocess.Popen([tool, "publish", path, "9", "30000"], stdout=subprocess.PIPE, text=True)
    old.stdout.readline()
    old.kill()
    old.wait()
    # a reader of the new launch (same nonce: same port, same shell) polls while the stale record is all there is ...
    rd = subprocess.Popen([tool, "read", path, "9", "5000"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    assert rd.poll() is None  # ... and does not take it
    new = subprocess.Popen([tool, "publish", path, "9", "3000"], stdout=subprocess.PIPE, text=True)
    try:
        out, _ = rd.communicate(timeout=10)
        assert rd.returncode == 0 and out.startswith("ok")
    finally:
        new.kill()
        new.wait()


def test_garbage_and_truncated_files_are_rejected(tool, tmp_path):
    path = str(tmp_path / "id")
    open(path, "wb").write(b"\x01" * 128)  # the bare 128-byte id of the round-2 format
    assert _read(tool, path, 1)[0] == 3
    open(path, "wb").write(os.urandom(4096))
    assert _read(tool, path, 1) == (3, "not a communicator-id record")


def test_nonce_is_shared_by_siblings_and_differs_between_launchers(tool):
    env = dict(os.environ, MASTER_PORT="29511")
    a = subprocess.run([tool, "nonce"], capture_output=True, text=True, env=env).stdout
    b = subprocess.run([tool, "nonce"], capture_output=True, text=True, env=env).stdout
    assert a == b  # same parent (this process), same port
    c = subprocess.run([tool, "nonce"], capture_output=True, text=True, env=dict(env, MASTER_PORT="29512")).stdout
    assert c != a
    d = subprocess.run(["bash", "-c", "%s nonce; true" % tool], capture_output=True, text=True, env=env).stdout  # another parent
    assert d != a
