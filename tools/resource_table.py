"""stdin: `make -C csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage); stdout: one line per kernel — VGPRs, AGPRs,
scratch bytes per lane, occupancy, LDS bytes per block — the table committed as profiles/rNN/k_kernel_resources.txt."""
import re, subprocess, sys
cur = None
rows = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                out = subprocess.run([tool, cur], capture_output=True, text=True).stdout.strip()
            except OSError:
                continue
            if out and not out.startswith("_Z"):
                # keep the template arguments of k_match2<...>, drop the parameter list
                depth, cut = 0, len(out)
                for i_, ch in enumerate(out):
                    depth += ch == "<"
                    depth -= ch == ">"
                    if ch == "(" and depth == 0:
                        cut = i_
                        break
                cur = out[:cut]
                break
        rows[cur] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, ln)
        if m and cur:
            rows[cur][key] = int(m.group(1))
print(f"{'kernel':86s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for k, v in sorted(rows.items()):
    if "o3s::kern" not in k and "_ZN3o3s4kern" not in k:
        continue
    print(f"{k[:86]:86s} {v.get('vgpr', 0):5d} {v.get('agpr', 0):5d} {v.get('scratch', 0):8d} {v.get('occ', 0):4d} {v.get('lds', 0):7d}")
