import os, sys, json
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import bench_joint
for trim in (True, False):
    r = bench_joint.run(128, 128, 300, 50, trim=trim)
    print(os.environ.get("E3D_SAMPLE_GRAPH"), "trim", trim, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("structure_s", "sequence_s", "total_s")}, flush=True)
