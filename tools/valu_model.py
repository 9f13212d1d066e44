"""VALU-issue model of a whole-trajectory kernel's step loop (bench.py's `roofline` with bound "valu").

The trajectory kernels keep the state in registers and draw the increments from the counter RNG, so a solve moves
2 * B * d * 4 bytes through HBM and is bound by what one SIMD can ISSUE: every vector instruction of the step loop
occupies the SIMD's VALU port for its issue cost. This tool

  1. compiles `torchsde_amd/csrc/trajectory.hip` to gfx950 assembly (device only; hipcc cross-compiles without a GPU),
  2. cuts the step loop of one kernel instantiation out of it (the loop header block and its latch blocks; the blocks of
     the output branch run once per requested output time, not once per step, and are left out),
  3. counts the loop's instructions by mnemonic, and
  4. prices the VALU ones with the per-instruction issue cycles measured on the device by tools/microbench_valu.hip
     (profiles/valu_rates.json; an instruction that was not measured costs the guide's 2 cycles per plain wave64 op and
     is listed under `unmeasured`).

    python tools/valu_model.py [--kernel SYMBOL_SUBSTRING] [--out profiles/valu_model_latest.json]

The result is stamped with the digest of the kernel sources; bench.py attaches it only if that digest is the one it runs.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEADLINE_SYMBOL = "_ZN4tsde17trajectory_kernelIfLi0ELi4ELb0ELb0EEEvNS_8TrajArgsIT_EE"


def affine_symbol(method, width=4, timed=False):
    """trajectory_kernel<float, METHOD, W, /*SENS=*/false, TIMED> (METHOD: include/torchsde_amd.h TSDE_TRAJ_*)."""
    return f"_ZN4tsde17trajectory_kernelIfLi{method}ELi{width}ELb0ELb{int(timed)}EEEvNS_8TrajArgsIT_EE"
DEFAULT_CYCLES = 2.0        # guides/MI355X_MICROARCH.md: one plain wave64 VALU instruction issues over 2 cycles (SIMD-32)


def device_asm(source="trajectory.hip"):
    """gfx950 assembly of one kernel source, compiled with the library's own flags."""
    csrc = os.path.join(ROOT, "torchsde_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                        "-S", os.path.join(csrc, source), "-o", out], check=True, stderr=subprocess.DEVNULL)
        with open(out) as fh:
            return fh.read()


def kernel_body(asm, symbol):
    start = asm.index("\n" + symbol + ":")
    end = asm.index("s_endpgm", start)
    return asm[start:end].splitlines()


_LABEL = re.compile(r"^(\.LBB\d+_\d+):")
_INSTR = re.compile(r"^\s+([a-z_0-9]+)\b")


def step_loop(lines):
    """Instruction mnemonics of the depth-1 loop's hot path: from the loop header, falling through every forward
    conditional branch (the branches to the output code, taken at the few steps that are output times) down to the
    back edge; plus the latch blocks the assembler placed BEFORE the header, if the back edge goes through them."""
    blocks, order, cur = {}, [], None
    notes = {}
    for ln in lines:
        m = _LABEL.match(ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            notes[cur] = ln
            continue
        if cur is None:
            continue
        m = _INSTR.match(ln)
        if m and not ln.lstrip().startswith((";", ".")):
            blocks[cur].append((m.group(1), ln.split()[-1]))
    headers = [b for b in order if "This Loop Header: Depth=1" in notes[b]]
    if len(headers) != 1:
        raise RuntimeError(f"expected one depth-1 loop, found {headers}")
    header = headers[0]
    tag = "Header=" + header.lstrip(".L")
    latch = [b for b in order[:order.index(header)] if tag in notes[b] and "Depth=1" in notes[b]]
    hot, walked, back = [], [], None
    for b in order[order.index(header):]:
        if b != header and not (tag in notes[b] and "Depth=1" in notes[b]):
            break
        walked.append(b)
        for ins, target in blocks[b]:
            hot.append(ins)
            if ins.startswith(("s_cbranch", "s_branch")) and (target == header or target in latch):
                back = target
                break
            if ins == "s_branch":
                back = target
                break
        if back is not None:
            break
    if back is None:
        raise RuntimeError("no back edge found below the loop header")
    if back in latch:
        for b in latch[latch.index(back):]:
            hot += [ins for ins, _ in blocks[b]]
    return hot, {"header": header, "walked": walked, "back_edge_to": back}


def _base(mnemonic):
    for suffix in ("_e32", "_e64", "_dpp", "_sdwa"):
        if mnemonic.endswith(suffix):
            return mnemonic[: -len(suffix)]
    return mnemonic


# Scalar operands. In isolation a VALU instruction that reads an SGPR issues every ~4.1 cycles (microbench rows v_bitop3_b32,
# v_xor_b32(sgpr), v_mul_f32(sgpr)); in the step loop -- where 37 of 99 instructions read one -- making the Philox round keys
# VGPR-resident changed nothing (profiles/r5e_vgpr_round_keys_experiment.txt): the limit is on back-to-back scalar reads.
# The loop is therefore priced with the ALL-VGPR issue cost of those instructions (the lower, defensible figure).
_IN_LOOP = {"v_bitop3_b32": "v_bitop3_b32(vgpr)", "v_mad_u64_u32": "v_mad_u64_u32(vgpr)"}

# instructions that were measured under another name of the same encoding class / rate
_ALIASES = {"v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_mov_b32": "v_not_b32", "v_and_b32": "v_xor_b32",
            "v_or_b32": "v_xor_b32", "v_sub_u32": "v_add_u32", "v_cmp_le_u32": "v_cmp_lt_u32",
            "v_cmp_gt_u32": "v_cmp_lt_u32", "v_fmac_f32": "v_fma_f32", "v_fmaak_f32": "v_fmamk_f32",
            "v_cvt_f32_i32": "v_cvt_f32_u32"}


def model(symbol=HEADLINE_SYMBOL, rates_path=None, source="trajectory.hip", elements_per_lane=4, asm=None):
    import bench
    rates_path = rates_path or os.path.join(ROOT, "profiles", "valu_rates.json")
    rates, rates_meta = {}, None
    if os.path.exists(rates_path):
        with open(rates_path) as fh:
            rec = json.load(fh)
        col = rec.get("column") or ("shader" if rec.get("ticks_are_shader_cycles") else "wall_2p4ghz")
        rates = {k: v[col] for k, v in rec["cycles"].items()}
        rates_meta = {"file": os.path.relpath(rates_path, ROOT), "column": col, "device": rec.get("device")}
    hot, where = step_loop(kernel_body(asm if asm is not None else device_asm(source), symbol))
    hist = collections.Counter(_base(i) for i in hot)
    valu = {k: n for k, n in hist.items() if k.startswith("v_")}
    other = {k: n for k, n in hist.items() if not k.startswith("v_")}
    cycles, unmeasured, plain = 0.0, [], 0.0
    for k, n in valu.items():
        key = _ALIASES.get(k, k)
        key = _IN_LOOP[key] if _IN_LOOP.get(key) in rates else key
        if key in rates:
            cycles += n * rates[key]
        else:
            cycles += n * DEFAULT_CYCLES
            unmeasured.append(k)
        plain += n * DEFAULT_CYCLES
    return {
        "csrc_sha": bench.csrc_digest(), "kernel_symbol": symbol, "blocks": where,
        "valu_instructions_per_wave_step": sum(valu.values()),
        "valu_histogram": dict(sorted(valu.items(), key=lambda kv: -kv[1])),
        "other_instructions": dict(sorted(other.items(), key=lambda kv: -kv[1])),
        "elements_per_lane_step": elements_per_lane,
        "issue_cycles_per_wave_step": cycles,
        "issue_cycles_per_wave_step_all_plain": plain,
        "unmeasured_priced_at_2_cycles": sorted(unmeasured),
        "rates": rates_meta,
        "is": "SIMD cycles the VALU port is occupied per wave and solver step = sum over the loop's vector instructions of "
              "count x measured issue cycles (tools/microbench_valu.hip); scalar, branch and memory instructions issue from "
              "other ports",
    }


def all_affine(rates_path=None):
    """The models of the five constant-coefficient affine kernels (Euler, Milstein Ito / Stratonovich, midpoint, SRK; one
    16-byte group per lane), from one compilation."""
    import bench
    asm = device_asm()
    kernels = {}
    for method in range(7):        # TSDE_TRAJ_EULER .. TSDE_TRAJ_EULER_HEUN
        symbol = affine_symbol(method)
        kernels[symbol] = model(symbol, rates_path, asm=asm)
    return {"csrc_sha": bench.csrc_digest(), "kernels": kernels}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_model_latest.json"))
    args = ap.parse_args()
    rec = all_affine(args.rates)
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
        fh.write("\n")
    for symbol, m in rec["kernels"].items():
        print(f"{symbol}: {m['valu_instructions_per_wave_step']} VALU instructions, "
              f"{m['issue_cycles_per_wave_step']:.1f} issue cycles per wave-step"
              + (f" (unmeasured, priced at 2 cycles: {m['unmeasured_priced_at_2_cycles']})"
                 if m["unmeasured_priced_at_2_cycles"] else ""))


if __name__ == "__main__":
    main()
