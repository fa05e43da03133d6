"""Issue-time mix of a kernel's hottest loop, from the gfx950 ISA (CPU only; hipcc -save-temps):
    python tools/isa_mix.py > profiles/r05_isa_mix.json
For every kernel named below: the opcode histogram of everything inside its loops (the voice loop of k_sources' Synth part, the stage
loop of k_band_chain), every VALU opcode priced with the issue time of its class measured on the part
(profiles/r03_issue_rate.txt, 8 waves per SIMD: f32 add / sub / mul / fma, mov, and / xor / ashr / add_u32 1.06 ns; min / max / med3 /
cmp / cndmask / floor / rndne / cvt / lshl / bfe / perm / SDWA / DPP forms / mul_lo / mul_hi and every PACKED f32 op 1.75 ns; f64
1.85 ns; rcp / rsq / sqrt / exp / log / sin / cos 3.4 ns) -> the average ns per VALU instruction bench.py multiplies the profiled
dynamic VALU count (SQ_INSTS_VALU) with, instead of pricing every instruction at the fastest class."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero".split()
KERNELS = {"k_sources<9u, 6>": "_ZN3tdk9k_sourcesILj9ELi6EEEv", "k_sources<14u, 4>": "_ZN3tdk9k_sourcesILj14ELi4EEEv",
           "k_band_chain<4, false>": "_ZN3tdk12k_band_chainILi4ELb0EEEv", "k_band_chain<4, true>": "_ZN3tdk12k_band_chainILi4ELb1EEEv",
           "k_band_chain<5, false>": "_ZN3tdk12k_band_chainILi5ELb0EEEv", "k_synth_affine": "_ZN3tdk14k_synth_affineE"}
FAST, SLOW, F64, TRANS = 1.06, 1.75, 1.85, 3.4


def price(op, line):
    if not op.startswith("v_") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_nop")):
        return None
    if re.search(r"_(f64|u64|i64|b64)\b", op) and not op.startswith("v_mov_b64") and not op.startswith("v_pk_"):
        return ("f64", F64)
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return ("trans", TRANS)
    if op.startswith("v_pk_") or "sdwa" in line or "dpp" in line or "row_" in line:
        return ("packed/sdwa/dpp", SLOW)
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_mov_b64", "v_and_b32", "v_or_b32", "v_xor_b32",
                      "v_ashrrev", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc", "v_accvgpr", "v_mac_f32", "v_madak", "v_madmk", "v_fmaak", "v_fmamk")):
        return ("fast", FAST)
    return ("slow", SLOW)


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-save-temps", "-c", os.path.join(ROOT, "termdaw_amd", "csrc", "kernels.hip"), "-o", os.path.join(d, "k.o")],
                              cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(os.path.join(d, "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    out = {"_note": __doc__.split("\n\n")[0].replace("\n", " ")}
    for name, sym in KERNELS.items():
        m = re.search(r"^(%s\w*):" % re.escape(sym), s, flags=re.M)
        if not m:
            continue
        body = s[m.end():s.index(".Lfunc_end", m.end())].splitlines()
        # depth of every line: the assembler comments say "in Loop: Header=BBx_y Depth=n" per basic block
        depth, rows = 0, []
        for l in body:
            t = l.strip()
            dm = re.search(r"Depth=(\d+)", t)
            if t.startswith(".LBB") or t.startswith("; %bb"):
                depth = int(dm.group(1)) if dm else 0
                if t.startswith(".LBB"):
                    continue
            if not t or t.startswith((";", ".")):
                continue
            rows.append((depth, t))
        # everything inside a loop (depth >= 1): the voice loop of the source kernels, the stage loop of the chain kernel with the
        # loops nested in it -- a launch's dynamic instruction count is these, over and over
        want = 1
        priced = [p for p in (price(t.split()[0], t) for dp, t in rows if dp >= want) if p]
        hist = collections.Counter(p[0] for p in priced)
        avg = sum(p[1] for p in priced) / len(priced)
        out[name] = {"loop_depth": want, "valu_in_loop": len(priced), "classes": dict(hist), "avg_ns_per_valu": round(avg, 4),
                     "ns": {"fast": FAST, "slow": SLOW, "f64": F64, "trans": TRANS}}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
