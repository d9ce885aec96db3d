"""Price the vector instructions of a render kernel's tracking step with the measured per-instruction issue costs of
profiles/r01_inst_rate.txt (ns per wave-instruction per SIMD at saturation, 8 waves per SIMD): how far above "every instruction costs
one v_fma" the kernel's instruction mix lies.   python scripts/isa_price.py [mangled-name-substring]
Needs /tmp/isa/vpk.s (hipcc -S --cuda-device-only of csrc/vp_kernels.hip, DEV build flags of scripts/kernel_resources.py)."""
import re, sys, collections
rate = {}
for l in open("profiles/r01_inst_rate.txt"):
    m = re.match(r"^(\S+)\s+waves/SIMD 8:.*-> ([\d.]+) ns per wave-inst per SIMD", l)
    if m: rate[m.group(1)] = float(m.group(2))   # the last block of the file wins (the repeated run)
base = rate["v_fma_f32"]
key = sys.argv[1] if len(sys.argv) > 1 else "render_kILi0ENS_10RngPhiloxRILi7EEELb1ELb0ELb0ELb1ELb0ELi0ELb0"
lines = open("/tmp/isa/vpk.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2vp8" + key) or (key in l and l.endswith(":") and not l.startswith("\t")))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(op):
    op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if op.startswith("v_cndmask"): return "v_cndmask", rate["v_cndmask_e64_vcc"]   # (the bare v_cndmask_b32 row of the table is a dependency artefact of that microbenchmark)
    if op in rate: return op, rate[op]
    for k in (op + "_e32",):
        if k in rate: return op, rate[k]
    if op.startswith(("v_fmac", "v_fmaak", "v_fmamk", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32")): return "f32 add/mul/fma", rate["v_mul_f32"] if "mul" in op or "add" in op or "sub" in op else base
    if op.startswith(("v_mad_u64_u32",)): return op, rate["v_mad_u64_u32"]
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_log", "v_exp")): return "transcendental", rate["v_rcp_f32"]
    if op.startswith("v_cmp"): return "v_cmp", rate["v_cmp_lt_f32"]
    if op.startswith("v_cndmask"): return "v_cndmask", rate["v_cndmask_e64_vcc"]
    if op.startswith(("v_cvt", "v_floor", "v_fract", "v_rndne", "v_min", "v_max", "v_med3", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_alignbit", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mad_u32", "v_div_", "v_ldexp", "v_frexp", "v_lshl_add_u64", "v_sub_u32", "v_subrev_u32", "v_and_or", "v_or3", "v_lshl_or", "v_add3", "v_xad")): return "conversions, min/max, shifts, integer multiplies, divide helpers", rate["v_floor_f32"]
    if op.startswith(("v_add_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_bitop3", "v_mov", "v_add_co", "v_addc", "v_not", "v_sub_co", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_mbcnt", "v_mov_b64")): return "integer add / logic / moves", rate["v_add_u32"]
    return "other:" + op, rate["v_floor_f32"]
tot = collections.Counter(); cost = collections.Counter(); n = 0
inloop = False
for l in lines[start:end]:
    t = l.strip().split(";")[0].strip()
    if not t or t.startswith("."): continue
    op = t.split()[0]
    if not op.startswith("v_"): continue
    c, r = cls(op)
    tot[c] += 1; cost[c] += r; n += 1
print(f"{key}: {n} vector instructions in the kernel; v_fma_f32 = {base:.3f} ns")
for c, k in tot.most_common():
    print(f"  {c:70s} {k:5d}  {cost[c] / k:.2f} ns each")
print(f"  mean {sum(cost.values()) / n:.3f} ns = {sum(cost.values()) / n / base:.2f} x v_fma_f32")
