"""Static VALU instruction-class mix of the hot sections of k_horizon<guess_constant, staged, fast stack>.

gfx950 issues wave64 VALU instructions at two rates (scripts/inst_rates.py, profiles/r03/inst_rates.json): the
"fast" class (v_fma_f32 / v_mul_f32 / v_add_f32 / v_sub_f32 / v_fmac_f32, v_mov_b32, v_and / v_or / v_xor,
v_add_u32 / v_sub_u32, v_cndmask_b32, shifts) goes every ~2.4 cycles per SIMD when its VGPR sources sit in different
register banks (register number mod 4; 4.1 cycles when they collide), everything else (conversions, v_perm_b32,
min / max / min3 / max3, compares, v_lshl_add_*, 64-bit integer, FP64) every ~4.15 cycles.  The roofline of bench.py
prices the kernel against the time its instructions need at those rates; this script counts, from the compiler's
assembly, how many instructions of each class one node step / leaf step / refill contains.

Sections are found from the hand-written load blocks: the node step starts at the first asm block with three
global_load_dwordx4 (hz_load_node) and ends at the second one (hz_load_prim), the leaf step runs from there to the
end of the kernel's main loop, the refill section is what precedes the node step inside that loop.
usage: python scripts/isa_class_mix.py   (needs hipcc; writes profiles/valu_class_mix.json)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAST = ("v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32",
        "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_cndmask_b32", "v_lshlrev_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_bfe_u32", "v_bfe_i32", "v_and_or_b32", "v_or3_b32", "v_mul_u32_u24", "v_mad_u32_u24")
KERNEL = "_ZN2hz9k_horizonILi2ELb0ELb1ELb0ELb0ELb0EEEvNS_13HorizonParamsE"     # <ALG_GUESS, !COUNT, STAGE, !NODELET, !LEVELSTACK, !LEFT>


def classify(lines):
    fast = slow = 0
    slow_names = {}
    for l in lines:
        t = l.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        if base in FAST:
            fast += 1
        else:
            slow += 1
            slow_names[base] = slow_names.get(base, 0) + 1
    return fast, slow, slow_names


def bank_census(lines):
    """VGPR bank (register number mod 4) collisions among the sources of the fast-class instructions: a fast
    instruction whose sources share a bank issues in ~4.1 instead of ~2.4 cycles (profiles/r03/inst_rates.json); hipcc
    assigns registers without regard to banks."""
    st = {"fast_instructions": 0, "one_vgpr_source": 0, "all_sources_in_different_banks": 0, "two_sources_share_a_bank": 0,
          "all_sources_in_one_bank": 0, "same_register_twice": 0}
    for l in lines:
        t = l.strip().split(None, 1)
        if not t or not t[0].startswith("v_"):
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        if base not in FAST:
            continue
        ops = [o.strip() for o in t[1].split(",")] if len(t) > 1 else []
        regs = [int(re.search(r"v(\d+)", o).group(1)) for o in ops[1:] if re.fullmatch(r"-?\|?v\d+\|?", o)]
        if base in ("v_fmac_f32", "v_mac_f32") and re.fullmatch(r"v\d+", ops[0]):
            regs.append(int(ops[0][1:]))                     # the destination is the third source
        st["fast_instructions"] += 1
        banks = [r % 4 for r in regs]
        if len(regs) <= 1:
            st["one_vgpr_source"] += 1
        elif len(set(regs)) < len(regs):
            st["same_register_twice"] += 1
        elif len(set(banks)) == len(banks):
            st["all_sources_in_different_banks"] += 1
        elif len(set(banks)) == 1 and len(banks) >= 3:
            st["all_sources_in_one_bank"] += 1
        else:
            st["two_sources_share_a_bank"] += 1
    return st


def main():
    src = os.path.join(ROOT, "horayzon_amd", "csrc", "hz_horizon.hip")
    out_s = "/tmp/hz_horizon_classmix.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out_s, src], stderr=subprocess.DEVNULL)
    text = open(out_s).read().split("\n")
    a = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(text)) if text[i].startswith(".Lfunc_end"))
    k = text[a:b]
    # the asm blocks and how many 16 B loads each holds
    blocks, i = [], 0
    while i < len(k):
        if "#ASMSTART" in k[i]:
            j = next(q for q in range(i, len(k)) if "#ASMEND" in k[q])
            blocks.append((i, j, sum("global_load_dwordx4" in x for x in k[i:j])))
            i = j
        i += 1
    node = next(x for x in blocks if x[2] == 2)                      # 32 B node: two 16 B loads (three for the 48 B node of early round 4, four before)
    leaf = next(x for x in blocks if x[2] == 3 and x[0] > node[0])   # 48 B leaf record
    # main loop: the innermost loop header before the node block ... the last backward branch after the leaf block
    hdr = max(i for i in range(node[0]) if "=>This Inner Loop Header" in k[i] or "Inner Loop Header" in k[i])
    outer = max(i for i in range(hdr) if "Loop Header" in k[i] and i != hdr) if any("Loop Header" in k[i] for i in range(hdr)) else hdr
    # the leaf step's code follows the leaf load in program order; other blocks of the outer loop are laid out behind
    # it, so it is cut after as many VALU instructions as the calibrated model says one leaf step has, minus the
    # ~30 of the loop top (votes, leaf queue) that every iteration shares (profiles/valu_model.json)
    n_leaf = 187
    try:
        n_leaf = int(round(json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))["leaf_iter"])) - 30
    except Exception:
        pass
    cnt, leaf_end = 0, len(k)
    for i in range(leaf[1], len(k)):
        t = k[i].strip().split()
        if t and t[0].startswith("v_"):
            cnt += 1
            if cnt >= n_leaf:
                leaf_end = i + 1
                break
    sec = {"node_step": k[node[1]:leaf[0]], "leaf_step": k[leaf[1]:leaf_end], "refill_and_loop_overhead": k[outer:hdr]}
    res = {"kernel_asm_sha": bench.kernel_asm_sha(), "classes": "fast: " + ", ".join(FAST)}
    for name, lines in sec.items():
        f, s, names = classify(lines)
        res[name] = {"fast": f, "slow": s, "fast_fraction": f / max(f + s, 1), "slow_breakdown": names,
                     "vgpr_bank_census_of_fast_instructions": bank_census(lines)}
    with open(os.path.join(ROOT, "profiles", "valu_class_mix.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k_: (v if not isinstance(v, dict) else {q: v[q] for q in ("fast", "slow", "fast_fraction")}) for k_, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
