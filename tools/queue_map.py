#!/usr/bin/env python3
"""Which hardware queue (rocprofv3's queue id) ran which kernel chain of the last proof in a kernel trace: G2 MSM, witness map, the G1 jobs (told apart by their
k_msm_recode / k_glv_split launches in time order).   python tools/queue_map.py <results.db>"""
import subprocess, sys, os, collections
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "timeline.py"), sys.argv[1], "150"], capture_output=True, text=True).stdout
rows = []
for ln in out.splitlines():
    if ln.startswith("#") or ln.startswith("kernel") or not ln.strip():
        continue
    f = ln.rsplit(None, 6)
    if len(f) != 7:
        continue
    rows.append((f[0], int(f[1]), float(f[2]), float(f[3])))
chains = collections.OrderedDict()
for name, q, start, dur in rows:
    if "G2" in name or "gls" in name:
        key = "G2"
    elif any(t in name for t in ("ntt", "spmv", "qap", "from_mont")):
        key = "wm"
    elif name.startswith("k_msm") or "glv" in name:
        key = "G1"
    else:
        key = "other"
    chains.setdefault((key, q), []).append((start, start + dur))
for (key, q), v in sorted(chains.items(), key=lambda kv: min(s for s, _ in kv[1])):
    print(f"{key:6s} queue {q}: {len(v):3d} launches, {min(s for s, _ in v):8.1f} .. {max(e for _, e in v):8.1f} us")
