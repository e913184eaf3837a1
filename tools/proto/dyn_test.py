import sys, time
sys.path.insert(0, '/root/repo')
from pcl_tracking_amd import scene, tracker
P = int(sys.argv[1])
model, cloud = scene.make_model(2048), scene.make_scene(50000)
t = tracker.make_reference_tracker(particle_num=P, seed=1)
t.setReferenceCloud(model); t.setTrans(scene.initial_trans()); t.setInputCloud(cloud)
for i in range(3):
    t.compute(); t.synchronize(); print("frame", i, "ok", flush=True)
print(t.getResult())
