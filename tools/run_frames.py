"""Run a few tracked frames of BASELINE configs[1] (for rocprofv3 passes).  python3 tools/run_frames.py [frames] [P] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcl_tracking_amd import scene, tracker  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
model = scene.make_model(2048)
cloud = scene.make_scene(N, mode="organized" if N == 307200 else "voxel")
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model)
t.setTrans(scene.initial_trans())
t.setInputCloud(cloud)
for i in range(frames):
    t.compute()
t.synchronize()
r = t.getResult()
print("result", [round(float(r[k]), 5) for k in ("x", "y", "z", "roll", "pitch", "yaw")])
