"""Dev aid: the copies and table kernels of the LAST decode call of a `rocprofv3 --kernel-trace --memory-copy-trace` run of
scripts/trace_decode.py, on one time axis.     python scripts/copy_timeline.py <dir with *_memory_copy_trace.csv / *_kernel_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
cp, kn = rows("*memory_copy_trace.csv"), rows("*kernel_trace.csv")
ev = []
for r in cp:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")), int(r.get("Bytes", r.get("Size", 0) or 0))))
for r in kn:
    nm = r["Kernel_Name"]
    if "tab_kernel" in nm or "copyBuffer" in nm or "scatter" in nm or "zero_dead" in nm:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel " + nm.split("<")[0].split("(")[0][-28:], 0))
ev.sort()
# the last call: starts at the last tab_kernel that follows a gap of more than 1 ms without tab kernels
tabs = [e for e in ev if "tab_kernel" in e[2]]
start = tabs[0][0]
for a, b in zip(tabs, tabs[1:]):
    if b[0] - a[1] > 1_000_000: start = b[0]
t0 = start
last = [e for e in ev if e[0] >= t0 - 300_000]
busy = 0
for s, e, nm, by in last:
    dur = (e - s) / 1e3
    rate = f"{by / max(e - s, 1):6.1f} GB/s" if by > (1 << 16) else ""
    print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  {dur:8.1f} us  {nm:40s} {by:>10d} {rate}")
big = [(s, e, by) for s, e, nm, by in last if nm.startswith("copy") and by > (1 << 20)]
if big:
    tot = sum(b for _, _, b in big)
    print(f"large copies: {tot / 1e6:.1f} MB from {(big[0][0] - t0) / 1e6:.3f} to {(big[-1][1] - t0) / 1e6:.3f} ms = {tot / (big[-1][1] - big[0][0]):.1f} GB/s; busy {sum(e - s for s, e, _ in big) / 1e6:.3f} ms")
