#!/usr/bin/env python3
"""Static instruction counts of one kernel by source line (no GPU needed).

    python tools/isa_lines.py render_persistent_kernelILi2ELb0ELb0 [--top 60] [--flags "-DX"]

Compiles ct_kernels.hip for gfx950 with the product's flags plus -gline-tables-only, walks the assembly of the kernel whose
mangled name contains the given string, and attributes every instruction to the innermost source line of its `.loc`.
Prints the lines with the most VALU instructions (file:line, VALU / SALU / LDS / memory counts, source text).  Static
counts: a line inside a loop counts once; read it together with the phase statistics of `tools/march_stats.py`.
"""
import argparse, collections, pathlib, re, subprocess, tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "deepestscatter_amd" / "csrc" / "ct_kernels.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
         "--cuda-device-only", "-gline-tables-only", "-S", f"-I{ROOT / 'include'}"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--flags", default="")
    ap.add_argument("--asm", default=None, help="use this assembly file instead of compiling")
    a = ap.parse_args()
    if a.asm:
        text = pathlib.Path(a.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = pathlib.Path(d) / "k.s"
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *a.flags.split(), "-o", str(out), str(SRC)], check=True, stderr=subprocess.DEVNULL)
            text = out.read_text()
    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', text, re.M):
        files[int(m.group(1))] = (pathlib.Path(m.group(2)) / m.group(3))
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and a.kernel in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    cur = (0, 0)
    counts = collections.defaultdict(lambda: collections.Counter())
    total = collections.Counter()
    for l in lines[start:end + 1]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
            k = "wait"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k = "mem"
        else:
            k = "other"
        counts[cur][k] += 1
        total[k] += 1
    print(f"kernel {lines[start].split(':')[0]}: {dict(total)}")
    by_file = collections.Counter()
    for (f, _), c in counts.items():
        by_file[files.get(f, f)] += c["valu"]
    print("VALU by file:", {str(pathlib.Path(str(k)).name): v for k, v in by_file.most_common()})
    src_cache = {}
    for (f, ln), c in sorted(counts.items(), key=lambda kv: -kv[1]["valu"])[:a.top]:
        p = files.get(f)
        if p not in src_cache:
            try:
                src_cache[p] = pathlib.Path(p).read_text().split("\n")
            except Exception:
                src_cache[p] = []
        srcl = src_cache[p][ln - 1].strip()[:110] if 0 < ln <= len(src_cache[p]) else ""
        print(f"{pathlib.Path(str(p)).name}:{ln:5d}  v{c['valu']:4d} s{c['salu']:3d} l{c['lds']:2d} m{c['mem']:2d}  {srcl}")


if __name__ == "__main__":
    main()
