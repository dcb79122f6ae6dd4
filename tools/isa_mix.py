"""Static VALU instruction mix of every kernel of libptamd.so by issue class (here, no GPU needed):

    python tools/isa_mix.py profiles/round3/isa_mix.json

Classes and cycles per wave64 instruction are the ones tools/micro/valu_issue.hip measured on the MI355X (profiles/round2/
r2c_valu_issue.md, W = 8 waves per SIMD): full rate ~2 cycles (v_fma/mul/add/sub/fmac_f32, and/or/xor, shifts, add/sub_u32, mov),
half rate ~4 cycles (everything else: min/max(3), compares, v_cndmask, conversions, bfe/perm/bfi, VOP3 integer ops, packed f32),
quarter rate ~8 cycles (v_rcp/rsq/sqrt/exp/log/sin/cos_f32).  avg_cycles = the static mix's mean: tools/traffic_json.py multiplies
it with the SQ_INSTS_VALU counter of a launch to price the vector issue port (valu_issue_frac)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opencl-path-tracer_amd", "csrc")
FULL = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_add_co_u32", "v_addc_co_u32",
        "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
# <any hit, LEVELS> (round 6; rounds 2-5: <any hit, two level[, start states]>): the names without a second argument are the instantiations for scenes that are one world-space tree (the headline)
KERNELS = {"k_traceILb1ELi0E": "k_trace<true>", "k_traceILb0ELi0E": "k_trace<false>", "k_traceILb1ELi1E": "k_trace<true, two-level>",
           "k_traceILb0ELi1E": "k_trace<false, two-level>", "k_traceILb1ELi2E": "k_trace<true, general>", "k_traceILb0ELi2E": "k_trace<false, general>",
           "k_trace_packetILb0ELb0": "k_trace_packet<false>",
           "k_trace_packetILb0ELb1": "k_trace_packet<false, two-level>", "k_trace_multiILi4ELb0": "k_trace_multi<4>",
           "k_trace_multiILi4ELb1": "k_trace_multi<4, two-level>", "k_shadeILb0ELb0ELb0": "k_shade<false>",
           "k_shadeILb0ELb1ELb0": "k_shade<false, general>", "5k_genE": "k_gen", "k_fold_planes": "k_fold_planes", "k_resolve": "k_resolve"}


def main():
    out = sys.argv[1]
    from_build = ["-O3", "-std=c++17", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"]
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *from_build, "-S", "--cuda-device-only", "ptamd.hip", "-o", "-"], cwd=CSRC,
                         check=True, capture_output=True, text=True).stdout
    res, hot, cur, depth = {}, {}, None, 0
    for line in asm.splitlines():
        m = re.match(r"^(_ZN3ptd\S+):", line)
        if m:
            cur = next((v for k, v in KERNELS.items() if k in m.group(1)), None)
            depth = 0
            if cur:
                res[cur], hot[cur] = collections.Counter(), collections.Counter()
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if cur:
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):  # a basic block: the compiler notes the loop it belongs to
                d = re.search(r"Depth=(\d+)", line)
                depth = int(d.group(1)) if d else 0
            m = re.match(r"\s+(v_\w+)", line)
            if m:
                op = re.sub(r"_e32|_e64|_sdwa|_dpp", "", m.group(1))
                res[cur][op] += 1
                hot[cur][(op, depth)] += 1
    table = {}

    def mix(c):
        full = sum(v for op, v in c.items() if op in FULL)
        quarter = sum(v for op, v in c.items() if op in QUARTER)
        total = sum(c.values())
        return total, full, total - full - quarter, quarter
    for k, c in res.items():
        total, full, half, quarter = mix(c)
        # the innermost loops are where the instructions are executed: the same mix over the blocks of the deepest loop level
        # that holds at least a fifth of the kernel's vector instructions
        levels = sorted({d for (_, d) in hot[k]}, reverse=True)
        inner = collections.Counter()
        for lv in levels:
            for (op, d), v in hot[k].items():
                if d == lv:
                    inner[op] += v
            if sum(inner.values()) >= 0.2 * total:
                break
        it, ifu, ih, iq = mix(inner)
        table[k] = {"valu_static": total, "full_rate": full, "half_rate": half, "quarter_rate": quarter,
                    "avg_cycles": round((2 * full + 4 * half + 8 * quarter) / max(total, 1), 3),
                    "inner_loops": {"valu_static": it, "full_rate": ifu, "half_rate": ih, "quarter_rate": iq,
                                    "avg_cycles": round((2 * ifu + 4 * ih + 8 * iq) / max(it, 1), 3)},
                    "top": dict(c.most_common(8))}
    json.dump({"source": "static instruction counts of the gfx950 code object (whole kernel), classes from profiles/round2/r2c_valu_issue.md",
               "cycles": {"full_rate": 2, "half_rate": 4, "quarter_rate": 8}, "kernels": table}, open(out, "w"), indent=1)
    for k, v in table.items():
        i = v["inner_loops"]
        print(f"{k:26s} {v['valu_static']:5d} VALU: {v['full_rate']:4d} full, {v['half_rate']:4d} half, {v['quarter_rate']:3d} quarter -> {v['avg_cycles']:.2f} cycles / instruction;"
              f"  inner loops {i['valu_static']:4d}: {i['full_rate']:4d} / {i['half_rate']:4d} / {i['quarter_rate']:3d} -> {i['avg_cycles']:.2f}")


if __name__ == "__main__":
    main()
