# Needs the probe build of the library (its switch does not exist in the default one; scri_amd/csrc/env.h):
#   make -C scri_amd/csrc PROBES=1 && export SCRI_AMD_LIB_PATH=$PWD/scri_amd/libscri_amd_probes.so
"""Timeline of the evaluating product's workgroups (debug trace: SCRI_AMD_GEMM_EVAL_TRACE=<file>): do the epilogues of the
workgroups that share a CU coincide?"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 5)
a = a[a[:, 4] > 0]
hw = (a[:, 0] & 0xFFFFFFFF).astype(np.int64)
xcc = (a[:, 0] >> 32).astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
wave = hw & 0xF
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0, t1, t2 = [a[:, k].astype(np.int64) for k in (2, 3, 4)]
base = t0.min()
print("blocks", len(a), "distinct CUs", len(np.unique(key)), "K loop (cycles) median", np.median(t1 - t0), "epilogue median", np.median(t2 - t1),
      "p90", np.percentile(t2 - t1, 90), "span", (t2.max() - base))
# overlap: for every CU, fraction of epilogue time during which another epilogue of the same CU is running
tot = ov = 0
for k in np.unique(key)[:64]:
    m = key == k
    s, e = t1[m], t2[m]
    order = np.argsort(s)
    s, e = s[order], e[order]
    for i in range(len(s)):
        tot += e[i] - s[i]
        for j in range(max(0, i - 4), min(len(s), i + 5)):
            if j != i:
                ov += max(0, min(e[i], e[j]) - max(s[i], s[j]))
print("epilogue time overlapped by another epilogue of the same CU: %.2f" % (ov / tot))
k = np.unique(key)[3]
m = key == k
o = np.argsort(t0[m])
print("one CU, first 12 blocks (start, K end, exit; relative, kcycles):")
for r in o[:12]:
    print("  wave slot %2d simd %d: %8.1f %8.1f %8.1f   tile %d,%d" % (wave[m][r], simd[m][r], (t0[m][r] - base) / 1e3, (t1[m][r] - base) / 1e3, (t2[m][r] - base) / 1e3,
                                                             a[m][r, 1] & 0xFFFFFFFF, a[m][r, 1] >> 32))
# K loop duration by column panel: is the half-empty last panel (its waves wn = 1 skip their products) any quicker?
bn = (a[:, 1] >> 32).astype(np.int64)
for p in sorted(set([0, 1, int(bn.max()) - 1, int(bn.max())])):
    sel = bn == p
    print("panel %2d: K loop median %8.0f, epilogue median %8.0f (%d tiles)" % (p, np.median((t1 - t0)[sel]), np.median((t2 - t1)[sel]), sel.sum()))
