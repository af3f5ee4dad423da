#!/usr/bin/env python3
"""Timeline of the DMA-fed factor kernel's workgroups (diagnostics build: profiles/tools/mkwx.sh stamp -DWX_STAMP): per 16x16 patch of
lines when lane 0 got its first and last row, how many of the courier's deliveries found the value missing and how often it polled again.
usage: ILUPP_HIP_LIBRARY=profiles/tools/lib_stamp.so [ILUPP_WD_MODE=1] wd_timeline.py GRID"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import matgen
from ilupp_amd import _native
dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256").split(",")]
dims = dims * 3 if len(dims) == 1 else dims
d, i, p = matgen.poisson3d(*dims)
n = p.shape[0] - 1
dev = torch.device("cuda", 0)
td, ti, tp = (torch.from_numpy(a).to(dev) for a in (d, i, p))
for rep in range(3):
    P = _native.ILU0Preconditioner_device(td.data_ptr(), ti.data_ptr(), tp.data_ptr(), n, True)
t = P.timings()
buf = (ctypes.c_ulonglong * (4096 * 4))()
assert _native.lib().ilupp_hip_debug_wf_timeline(buf) == 0
wbuf = (ctypes.c_ulonglong * (4096 * 16))()
assert _native.lib().ilupp_hip_debug_wf_wait(wbuf) == 0
Ty, Tz = dims[1] // 16, dims[2] // int(os.environ.get('ILUPP_TILE_TZ', '16'))
nt = Ty * Tz
a = np.array(buf[:nt * 4], dtype=np.float64).reshape(nt, 4)
w = np.array(wbuf[:nt * 16], dtype=np.float64).reshape(nt, 16)
t0 = a[:, 0].min()
a = (a - t0) / 100.0      # us
print("factor kernel %.1f us; tiles %d x %d (by ticket)" % (1e3 * t["numeric_kernel_ms"], Ty, Tz))
print("first row of lane 0 (us):")
for z in range(Tz):
    print(" ".join("%6.1f" % a[z * Ty + y, 1] for y in range(Ty)))
print("first -> last row of lane 0 (us), i.e. %d steps:" % (dims[0] - 1))
for z in range(Tz):
    print(" ".join("%6.1f" % (a[z * Ty + y, 2] - a[z * Ty + y, 1]) for y in range(Ty)))
print("deliveries that found the value missing, of %d steps:" % int(w[0, 15]))
for z in range(Tz):
    print(" ".join("%6d" % w[z * Ty + y, 13 if os.environ.get("ILUPP_NO_WA") else 14] for y in range(Ty)))
print("polls repeated:")
for z in range(Tz):
    print(" ".join("%6d" % w[z * Ty + y, 4] for y in range(Ty)))
dur = a[:, 2] - a[:, 1]
if not os.environ.get("ILUPP_NO_WA"):
    npf = w[:, 9]; lead = np.array(wbuf[:nt * 16], dtype=np.uint64).reshape(nt, 16)[:, 11].astype(np.int64)
    print("prefetchers: %d blocks of eight steps asked for in all (%.1f per tile; a tile has %d), %d scans; a block was asked for %.1f steps ahead of its tile's progress on average" %
          (npf.sum(), npf.sum() / nt, (dims[0] + 30) // 8, (np.array(wbuf[:nt * 16], dtype=np.uint64).reshape(nt, 16)[:, 10] & np.uint64(0xffffffff)).sum(), lead.sum() / max(1.0, npf.sum())))
    print("blocks finished by the prefetcher waves: %d" % (np.array(wbuf[:nt * 16], dtype=np.uint64).reshape(nt, 16)[:, 10] >> np.uint64(32)).sum())
    print("blocks asked for by the prefetcher of each workgroup:")
    for z in range(Tz):
        print(" ".join("%4d" % npf[z * Ty + y] for y in range(Ty)))
if not os.environ.get("ILUPP_NO_WA"):
    hw = np.array(wbuf[:nt * 16], dtype=np.uint64).reshape(nt, 16)[:, 8]
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
    cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    phys = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    from collections import defaultdict
    byc = defaultdict(list)
    for k in range(nt): byc[int(phys[k])].append(k)
    print("XCD of each tile:")
    for z in range(Tz):
        print(" ".join("%d" % xcc[z * Ty + y] for y in range(Ty)))
    sizes = sorted(len(v) for v in byc.values())
    print("physical CUs used: %d; tiles per CU: min %d max %d; xcc of tiles 0..15: %s" % (len(byc), sizes[0], sizes[-1], " ".join(str(int(v)) for v in xcc[:16])))
    ov = []
    for c, ks in byc.items():
        for i in range(len(ks)):
            for j in range(i + 1, len(ks)):
                lo = max(a[ks[i], 1], a[ks[j], 1]); hi = min(a[ks[i], 2], a[ks[j], 2])
                ov.append(max(0.0, hi - lo))
    if ov: print("tiles sharing a CU: %d pairs, their rows overlap in time for %.1f us on average (a tile runs %.1f us); examples: %s" % (len(ov), float(np.mean(ov)), float(np.mean(dur)), " ".join(str(v) for v in list(byc.values())[:6])))
cyc = w[:, 12]
if os.environ.get("ILUPP_NO_WA"):
    wi = np.array(wbuf[:nt * 16], dtype=np.uint64).reshape(nt, 16)
    ld0_bar = (wi[:, 5] & np.uint64(0xffffffff)).astype(np.float64)
    ld0_win = (wi[:, 5] >> np.uint64(32)).astype(np.float64)
    ld1_bar = (wi[:, 6] & np.uint64(0xffffffff)).astype(np.float64)
    ld1_iss = (wi[:, 6] >> np.uint64(32)).astype(np.float64)
    def sh(v, k):
        return "%.2f" % (v[k] / cyc[k])
    for k in (0, nt // 2 + Ty // 2, nt - 1):
        print("tile %d: share of its life (cycles %d): consumers at barriers %s; loaders at barriers %s %s %s %s, loader 0 waiting for its windows %s; poller at barriers %s, delivering %s; exporter at barriers %s, delivering %s" %
              (k, cyc[k], " ".join(sh(w[:, q], k) for q in range(4)), sh(ld0_bar, k), sh(ld1_bar, k), sh(w[:, 7], k), sh(w[:, 8], k), sh(ld0_win, k) + " (loader 1 issuing: " + sh(ld1_iss, k) + ")",
               sh(w[:, 11], k), sh(w[:, 14], k), sh(w[:, 9], k), sh(w[:, 10], k)))
else:
    def sh(v, k):
        return "%.2f" % (v[k] / cyc[k])
    for k in (0, nt // 2 + Ty // 2, nt - 1):
        print("tile %d: share of its life (cycles %d): consumer waves waiting for counters %s (wave 0: %d of %d steps); loader 0: waiting for its wave %s, issuing %s, waiting for its windows %s; poller: %d values missing at first look, %d polls repeated" %
              (k, cyc[k], " ".join(sh(w[:, q], k) for q in range(4)), w[k, 13], w[k, 15], sh(w[:, 5], k), sh(w[:, 6], k), sh(w[:, 7], k), w[k, 14], w[k, 4]))
life = (a[:, 3] - a[:, 0])
print("shader clock over each workgroup's life: median %.2f GHz (min %.2f max %.2f)" % tuple(f(w[:, 12] / (life * 1e3)) for f in (np.median, np.min, np.max)))
print("us per step while a tile runs: median %.3f min %.3f max %.3f; entry of the last tile %.1f us, end of the last row %.1f us" %
      (np.median(dur) / (dims[0] - 1), dur.min() / (dims[0] - 1), dur.max() / (dims[0] - 1), a[:, 1].max(), a[:, 2].max()))
