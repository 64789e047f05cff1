#!/usr/bin/env python3
"""glimmer-mg -r (circular genome) on NC_000915.fna: the all-reference binary against glimmer-mg_gpu, which hands -r to the drop-in binary
(the reference's main() on the device-backed ICM_t: Score_All_Frames = 12 Frame_Score calls per SEQUENCE, each one launch over the whole genome)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "golden", "data")
out = []
for name, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "glimmer-mg")), ("glimmer-mg_gpu -r", os.path.join(ROOT, "integration", "_build", "glimmer-mg_gpu"))):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([exe, "-r", "-m", os.path.join(D, "NC_000915.icm"), os.path.join(D, "NC_000915.fna"), "/tmp/circ_" + name.split()[0]],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append((name, best))
    print("%-20s %.3f s" % (name, best))
a, b = (open("/tmp/circ_%s.predict" % n.split()[0], "rb").read() for n, _ in out)
print("identical" if a == b else "DIFFERENT", a.count(b"orf"), "genes")
