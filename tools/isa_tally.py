#!/usr/bin/env python3
"""Static ISA of the compositing kernels' inner loops (gfx950), with an instruction tally per basic block.

    python tools/isa_tally.py [out.txt]

Compiles csrc/raster_fwd.hip and csrc/raster_bwd.hip to assembly with the library's own flags (hipcc -S,
device only; cross-compiles without a GPU), takes the two kernels the default bench runs --
raster_fwd_tile16_kernel<false> and raster_bwd_tile16_kernel<4, false> -- finds the innermost loop over the staged
splats (`for t < count`), and prints its basic blocks with, per block: VALU (of which transcendental / DPP +
permlane / v_cndmask), SALU, branches, s_waitcnt, LDS, VMEM, atomics.  The thread-trace decoder is absent from this
image; the static loop body is not.  What the blocks ARE (per splat / per pixel / butterfly) is annotated in
DESIGN.md section 4.1 / 4.2 from this listing."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-splatting-toolkit_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-munsafe-fp-atomics",
         "-fno-slp-vectorize", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(ins):
    op = ins.split()[0]
    c = {"valu": 0, "trans": 0, "dpp": 0, "cndmask": 0, "salu": 0, "branch": 0, "waitcnt": 0, "lds": 0, "vmem": 0,
         "atomic": 0, "readlane": 0}
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc"):
        c["branch"] = 1
    elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        c["waitcnt"] = 1
    elif op.startswith("s_"):
        c["salu"] = 1
    elif op.startswith("ds_"):
        c["lds"] = 1
    elif "atomic" in op:
        c["atomic"] = 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] = 1
    elif op.startswith("v_"):
        c["valu"] = 1
        if op.startswith(TRANS):
            c["trans"] = 1
        if "dpp" in ins or op.startswith("v_permlane"):
            c["dpp"] = 1
        if op.startswith("v_cndmask"):
            c["cndmask"] = 1
        if op.startswith(("v_readfirstlane", "v_readlane")):
            c["readlane"] = 1
    return c


def issue_cycles(ins):
    """Issue cost of one wave64 VALU instruction on a gfx950 SIMD, in cycles, as MEASURED by tools/exp/valubench2.hip
    (8 resident waves per SIMD, independent instructions; profiles/r05_valubench2.txt):
      8.2  transcendentals (v_exp / v_rcp / v_rsq / ...) and v_permlane{16,32}_swap
      4.2  v_min / v_max, every v_cmp, v_cndmask, anything with a DPP modifier, v_cvt, v_readfirstlane / v_readlane,
           the VOP3-only integer ops (v_lshl_add, v_bfe, v_med3, v_mad ...), v_pk_* (two lane-ops per lane), and ANY
           plain instruction with an SGPR source operand (v_mul_f32 v, s, v: 4.14 against 2.45 with two VGPRs)
      2.7  v_fma_f32 (VOP3)
      2.4  v_add / v_sub / v_mul (e32 and e64) / v_fmac / v_fmamk / v_fmaak / v_mov / v_and / v_or / v_xor / shifts /
           v_add_u32
    0 for everything that is not a VALU instruction."""
    op = ins.split()[0]
    if not op.startswith("v_"):
        return 0.0
    if op.startswith(TRANS) or op.startswith("v_permlane"):
        return 8.2
    ops = ins[len(op):]
    sgpr_src = bool(re.search(r",\s*-?\|?s(\d+|\[)", ops)) or (", vcc" in ops and not op.startswith("v_cndmask"))
    if ("dpp" in ins or "quad_perm" in ins or "row_" in ins or op.startswith(
            ("v_min_", "v_max_", "v_cmp", "v_cndmask", "v_cvt_", "v_readfirstlane", "v_readlane", "v_lshl_add", "v_bfe",
             "v_med3", "v_mad_", "v_pk_", "v_bfi", "v_add3", "v_lshl_or", "v_and_or", "v_or3", "v_mbcnt", "v_mul_lo",
             "v_mul_hi", "v_ldexp", "v_frexp", "v_trunc", "v_floor", "v_rndne", "v_fract"))) or sgpr_src:
        return 4.2
    if op.startswith("v_fma_f32"):
        return 2.7
    return 2.4


def kernel_text(asm, needle):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and ": " in l and "@" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def innermost_loop(lines):
    """-> (header label, [block lines]) of the deepest loop that reads LDS (the walk over the staged splats)."""
    best = None
    for i, l in enumerate(lines):
        if not l.startswith(".LBB"):
            continue
        ctx, k = l, i + 1
        while k < len(lines) and lines[k].lstrip().startswith(";") and k < i + 6:  # the loop comments span lines
            ctx += lines[k]
            k += 1
        m = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", ctx)
        if not m:
            continue
        label, depth = l.split(":")[0], int(m.group(1))
        tag = "Header=%s Depth=%d" % (label[2:], depth)
        last = max([k for k, x in enumerate(lines) if tag in x] + [i])
        k = last
        while k + 1 < len(lines) and not lines[k + 1].startswith(".LBB"):
            k += 1
        body = lines[i:k + 1]
        # the walk over the staged splats: reads them from LDS and evaluates exp(-sigma)
        if any(x.strip().startswith("ds_") for x in body) and any(x.strip().startswith("v_exp_f32") for x in body) and (
                best is None or depth > best[2] or (depth == best[2] and len(body) < len(best[1]))):
            best = (label, body, depth)
    return best[0], best[1]


def tally(body):
    blocks, cur, name = [], [], "(header)"
    for l in body:
        s = l.strip()
        if l.startswith(".LBB") or s.startswith("; %bb."):
            if cur:
                blocks.append((name, cur))
            name, cur = (l.split(":")[0] if l.startswith(".LBB") else s.split(":")[0][2:]), []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur.append(s.split(";")[0].strip())
    if cur:
        blocks.append((name, cur))
    return blocks


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for src, needle, title in (("raster_fwd.hip", "raster_fwd_tile16_kernelILb0E", "raster_fwd_tile16_kernel<false>"),
                               ("raster_bwd.hip", "raster_bwd_tile16_kernelILi4ELb0E", "raster_bwd_tile16_kernel<4, false>")):
        with tempfile.TemporaryDirectory() as d:
            s_path = os.path.join(d, "k.s")
            subprocess.run([hipcc] + FLAGS + ["-o", s_path, os.path.join(CSRC, src)], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm = open(s_path).read()
        lines = kernel_text(asm, needle)
        vg = re.search(needle + r"[^\n]*\.num_vgpr, (\d+)", asm)
        label, body = innermost_loop(lines)
        blocks = tally(body)
        out.write("=" * 110 + "\n%s: innermost LDS-reading loop %s, %d basic blocks, kernel VGPRs %s\n" % (
            title, label, len(blocks), vg.group(1) if vg else "?"))
        out.write("%-14s %5s %5s %5s %5s %5s %5s %5s %5s %5s %5s %7s\n" % (
            "block", "VALU", "trans", "dpp", "cndm", "SALU", "brnch", "wait", "LDS", "VMEM", "atom", "cycles"))
        tot = None
        for name, ins in blocks:
            c = {}
            for i in ins:
                for k, v in classify(i).items():
                    c[k] = c.get(k, 0) + v
            c["cycles"] = sum(issue_cycles(i) for i in ins)
            tot = c if tot is None else {k: tot[k] + c.get(k, 0) for k in tot}
            out.write("%-14s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %7.1f\n" % (
                name, c.get("valu", 0), c.get("trans", 0), c.get("dpp", 0), c.get("cndmask", 0), c.get("salu", 0),
                c.get("branch", 0), c.get("waitcnt", 0), c.get("lds", 0), c.get("vmem", 0), c.get("atomic", 0),
                c["cycles"]))
        out.write("%-14s %5d %5d %5d %5d %5d %5d %5d %5d %5d %5d %7.1f   (every block once; cycles = measured VALU "
                  "issue cost per wave-instruction, tools/exp/valubench2.hip)\n\n" % (
            "sum", tot["valu"], tot["trans"], tot["dpp"], tot["cndmask"], tot["salu"], tot["branch"], tot["waitcnt"],
            tot["lds"], tot["vmem"], tot["atomic"], tot["cycles"]))
        for name, ins in blocks:
            out.write("-- %s\n" % name)
            for i in ins:
                cyc = issue_cycles(i)
                out.write("    %-72s%s\n" % (i, ("  ; %.1f" % cyc) if cyc else ""))
        out.write("\n")


if __name__ == "__main__":
    main()
