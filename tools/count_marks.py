"""Static instruction accounting for one kernel (developer tool): compile the kernel's translation unit (csrc/k_*.hip) with -DPOLAR_MARKS -S
(--cuda-device-only) and give the .s file and the kernel's mangled name; counts the instructions between consecutive
`; MARK name` comments, by (from, to) pair, split into VALU / SALU / LDS / VMEM / other, and the scratch (spill) traffic.

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -mllvm -amdgpu-sched-strategy=iterative-maxocc \
          -DPOLAR_MARKS -S --cuda-device-only -o /tmp/k.s polardecoding_amd/csrc/k_fast2.hip
    python tools/count_marks.py /tmp/k.s _ZN5polar11k_scl_fast2IddLb1EEEvNS_9SclParamsE
"""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
    cur = "begin"
    seg = collections.defaultdict(lambda: collections.Counter())
    occ = collections.Counter()
    tot = collections.Counter()
    for l in lines[start + 1:end]:
        m = re.search(r"; MARK (\S+)", l)
        if m:
            nxt = m.group(1)
            occ[(cur, nxt)] += 1
            cur_key = nxt
            cur = cur_key
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_")):
            k = "vmem"
        elif op.startswith("scratch_"):
            k = "spill"
        else:
            k = "other"
        seg[cur][k] += 1
        tot[k] += 1
    print("total", dict(tot))
    print(f"{'after mark':28s} {'n':>4s} {'valu':>7s} {'salu':>6s} {'lds':>5s} {'vmem':>5s} {'spill':>6s}   (instructions following the mark until the next one, summed over its n copies)")
    cnt = collections.Counter()
    for (a, b), n in occ.items():
        cnt[b] += n
    for k in sorted(seg, key=lambda k: -sum(seg[k].values())):
        c = seg[k]
        print(f"{k:28s} {cnt.get(k, 1):4d} {c['valu']:7d} {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d} {c['spill']:6d}")


if __name__ == "__main__":
    main()
