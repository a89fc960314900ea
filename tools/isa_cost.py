#!/usr/bin/env python3
"""Static issue-cost estimate of a gfx950 kernel from hipcc -S output, with the per-instruction costs measured by
tools/microbench/valu_rates.hip on one MI355X (wave64, cycles of the SIMD's issue port at 2.4 GHz):
  fast 2.6  f32 add / sub / mul / fma / fmac (VGPR, inline-constant or literal operands, neg / abs modifiers), and / or / xor / not,
            add / sub u32, v_mov from a VGPR
  slow 4.3  the same with an SGPR operand; min / max / med3; v_cndmask; compares; shifts, alignbit, bfe, perm, lshl_add, add3, and_or, lshl_or;
            mul_lo, mul24, mad24; conversions; div_scale / div_fmas / div_fixup; DPP; readlane / writelane / readfirstlane
  trans 8.3 rcp, rsq, sqrt, exp, log
usage: isa_cost.py file.s kernel-name-substring [--blocks]
"""
import re, sys
FAST_OPS = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mul_legacy_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
            "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_accvgpr")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
def classify(line):
    t = line.split()
    if not t: return None
    op = t[0]
    if not op.startswith("v_"): return None
    if op.startswith(TRANS): return "trans"
    base = re.sub(r"_e(32|64)$", "", op)
    if base.endswith("_dpp") or base.endswith("_sdwa"): return "slow"
    if base in FAST_OPS:
        ops = " ".join(t[1:])
        srcs = ops.split(",")[1:]                     # operands after the destination
        if any(re.search(r"(^|[\s\-|])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", x.strip()) for x in srcs): return "slow"
        return "fast"
    return "slow"
COST = {"fast": 2.6, "slow": 4.3, "trans": 8.3}
def main():
    path, key = sys.argv[1], sys.argv[2]
    per_block = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
    out, blk, tot = {}, "entry", {"fast": 0, "slow": 0, "trans": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"): break
        s = l.strip()
        if s.startswith(".LBB") and s.split()[0].endswith(":"): blk = s.split(":")[0]; continue
        if not s or s.startswith(";") or s.startswith("."): continue
        op = s.split()[0]
        c = classify(s)
        k = c if c else ("salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop")) else "smem" if op.startswith(("s_load", "s_buffer")) else
                         "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else None)
        if k is None: continue
        tot[k] += 1
        out.setdefault(blk, {"fast": 0, "slow": 0, "trans": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0})[k] += 1
    cyc = lambda d: sum(d[k] * COST[k] for k in COST)
    print(f"{key}: VALU fast {tot['fast']} slow {tot['slow']} trans {tot['trans']} = {cyc(tot):.0f} issue cycles static; SALU {tot['salu']} SMEM {tot['smem']} VMEM {tot['vmem']} LDS {tot['lds']}")
    if per_block:
        for b, d in out.items():
            if d["fast"] + d["slow"] + d["trans"] >= 8: print(f"  {b:12s} fast {d['fast']:4d} slow {d['slow']:4d} trans {d['trans']:2d}  {cyc(d):7.0f} cycles   salu {d['salu']} smem {d['smem']} vmem {d['vmem']} lds {d['lds']}")
if __name__ == "__main__": main()
