"""Host-side logic of bench.py that needs no GPU: how many CPUs / GPUs the process may really use."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_usable_cpus_respects_the_affinity_mask():
    b = _bench()
    usable, cpu_count, affinity, quota = b.usable_cpus()
    assert 1 <= usable <= affinity <= max(cpu_count, affinity)
    if quota is not None:
        assert usable <= max(1, int(quota + 0.999))
    # a child pinned to one CPU must report one usable CPU whatever os.cpu_count() says
    one = sorted(os.sched_getaffinity(0))[0]
    code = ("import os,sys,importlib.util; os.sched_setaffinity(0,{%d}); "
            "s=importlib.util.spec_from_file_location('b', %r); m=importlib.util.module_from_spec(s); s.loader.exec_module(m); "
            "print(m.usable_cpus()[0])" % (one, os.path.join(ROOT, "bench.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert r.stdout.strip() == "1"


def test_gpu_count_does_not_load_torch_and_honours_visible_devices(monkeypatch):
    code = ("import sys,importlib.util; "
            "s=importlib.util.spec_from_file_location('b', %r); m=importlib.util.module_from_spec(s); s.loader.exec_module(m); "
            "n=m.visible_gpu_count(); print(n, 'torch' in sys.modules)" % os.path.join(ROOT, "bench.py"))
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env)
    n, torch_loaded = r.stdout.split()
    has_kfd = os.path.isdir("/sys/class/kfd/kfd/topology/nodes")
    assert int(n) >= 0
    if has_kfd:
        assert torch_loaded == "False"  # the launcher parent counts GPUs from sysfs alone
        env["HIP_VISIBLE_DEVICES"] = ""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env)
        assert r.stdout.split()[0] == "0"
