#!/usr/bin/env python3
"""Builds libvlr_hip.so (gfx950) in-tree: hipcc cross-compiles without a GPU.

    python vl-rlhf_amd/build_hip.py [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libvlr_hip.so")
SOURCES = ["api.cpp", "layers.cpp", "comm.cpp", "gemm.hip", "gemm128p.hip", "lora_dx.hip", "lora_rows.hip", "gemm256p.hip", "attention.hip", "elementwise.hip", "dpo_ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I/opt/rocm/include"]


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/vlr.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def kernel_digest(sources=("gemm256p.hip", "gemm_tilemap.h", "gemm.h", "common.h")):
    """digest of the translation unit of ONE kernel family (default: the dominant 256x256 GEMM, gemm256p.hip and the three headers it
    includes) + the compiler flags: what tools/pmc_traffic.sh records next to the counters it collects and bench.py compares before it
    quotes them - a change to another kernel's source does not make the GEMM's counters stale, a change to the GEMM's does."""
    h = hashlib.sha256()
    for name in sources:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def check_kloop_isa(asm_path):
    """gemm256p.hip: the steady-state K loop of every kernel (the branch-free innermost loop with exactly 64 v_mfma: 4 phases x 16)
    must hold only the hand-written counted waits: vmcnt(6) (classic body) or vmcnt(8) + vmcnt(6) (balanced phases).  Any other s_waitcnt vmcnt in there is hipcc guarding a
    register against an epilogue load it believes pending - it drains the LDS-DMA queue every K tile (-7 % on the TN kernel when it
    happened).  Raises with the kernel name and the offending waits."""
    import re
    text = open(asm_path).read()
    bad, seen = [], 0
    for m in re.finditer(r"^(_Z15gemm256p_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        targs = re.match(r"_Z15gemm256p_kernelILb(\d)ELb(\d)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)E", name)
        if targs and int(targs.group(3)) != 0:
            continue                                   # timing-ablation instantiations (VLR_GEMM_ABLATE)
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
        for i, l in enumerate(body):
            t = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l)
            if not t or t.group(1) not in labels or labels[t.group(1)] >= i:
                continue
            loop = body[labels[t.group(1)]:i]
            if sum("v_mfma" in x for x in loop) != 64 or any(x.startswith(".LBB") for x in loop[1:]):
                continue
            seen += 1
            waits = [x.strip() for x in loop if "s_waitcnt" in x and "vmcnt" in x]
            if waits not in (["s_waitcnt vmcnt(6)"], ["s_waitcnt vmcnt(8)", "s_waitcnt vmcnt(6)"]):      # classic / balanced-phase K loop
                bad.append((name, waits))
    if bad:
        raise RuntimeError("gemm256p.hip: stray vector-memory waits in the steady-state K loop:\n" + "\n".join(f"  {n}: {w}" for n, w in bad))
    if seen < 12:
        raise RuntimeError(f"gemm256p.hip: the ISA check found only {seen} steady-state K loops (expected one per kernel) - update check_kloop_isa")


def check_attn_fwd3_isa(asm_path):
    """attention.hip: attn_fwd3_kernel owns a[0:191] by inline asm (attn_fwd3_regs.h: O = a[0:127], the wave's Q fragments = a[128:191]).
    hipcc only knows these registers as clobbers of single statements: a compiler-generated v_accvgpr_* or MFMA touching them between two
    statements (an AGPR used as VGPR spill space) would corrupt them silently.  Every accumulator register below a192 that appears
    outside ;;#ASMSTART / ;;#ASMEND fails the build; so does a vector register spill."""
    import re
    text = open(asm_path).read()
    seen = 0
    for m in re.finditer(r"^(_Z16attn_fwd3_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seen += 1
        inside, bad = False, []
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                inside = True
            elif "#ASMEND" in line:
                inside = False
            elif not inside:
                code = line.split(";")[0]
                for r in re.finditer(r"\ba\[(\d+):\d+\]|\ba(\d+)\b", code):
                    if int(r.group(1) or r.group(2)) < 192:
                        bad.append(code.strip())
        if bad:
            raise RuntimeError(f"attention.hip: {name}: hipcc touches the asm-owned accumulator registers a[0:191]:\n  " + "\n  ".join(bad[:8]))
    meta = re.findall(r"\.name:\s+_Z16attn_fwd3_kernel\S*.*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    if any(int(x) for x in meta):
        raise RuntimeError("attention.hip: attn_fwd3_kernel spills vector registers")
    if seen < 2:
        raise RuntimeError(f"attention.hip: the ISA audit found {seen} attn_fwd3_kernel instantiations (expected 2) - update check_attn_fwd3_isa")


def build(force=False, verbose=True, defines=(), tag=""):
    """defines / tag: a diagnostics variant of the library (e.g. defines=("VLR_GEMM_TRACE",), tag="_trace" -> libvlr_hip_trace.so with
    its own object directory), loaded through VLR_LIB; the product build is the one with neither."""
    OBJ = os.path.join(HERE, "build" + tag)
    LIB = os.path.join(HERE, f"libvlr_hip{tag}.so")
    FLAGS = globals()["FLAGS"] + [f"-D{d}" for d in defines]
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "stamp")
    dg = _digest() + "".join(defines)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dg:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def cc(src):
        obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".hip"):
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if src in ("gemm256p.hip", "attention.hip"):
            cmd.append("-save-temps")          # keeps the gfx950 assembly (in OBJ) for check_kloop_isa / check_attn_fwd3_isa
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        # a kernel that touches scratch (spill, or a register array the compiler could not keep in VGPRs) is a 10-20x
        # performance bug on this path: refuse to build it
        import re
        for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, re.S):
            if int(m.group(2)) > 0 and not tag:      # (diagnostics variants may: they are not what is measured or shipped)
                raise RuntimeError(f"{src}: kernel {m.group(1)} uses {m.group(2)} B/lane of scratch")
        if src == "gemm256p.hip":
            check_kloop_isa(os.path.join(OBJ, "gemm256p-hip-amdgcn-amd-amdhsa-gfx950.s"))
        if src == "attention.hip":
            check_attn_fwd3_isa(os.path.join(OBJ, "attention-hip-amdgcn-amd-amdhsa-gfx950.s"))
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dg)
    if verbose:
        print(f"[build_hip] built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)")
    return LIB


if __name__ == "__main__":
    if "--define" in sys.argv:     # any A/B variant: build_hip.py --define NAME=VALUE [--define ...] --tag _suffix  -> libvlr_hip_suffix.so (VLR_LIB)
        defs = tuple(sys.argv[i + 1] for i, a_ in enumerate(sys.argv) if a_ == "--define")
        build(force="--force" in sys.argv, defines=defs, tag=sys.argv[sys.argv.index("--tag") + 1])
    elif "--trace" in sys.argv:      # tile-timeline diagnostics of the persistent GEMM (tools/gemm_tile_trace.py)
        build(force="--force" in sys.argv, defines=("VLR_GEMM_TRACE",), tag="_trace")
    elif "--classic" in sys.argv:  # the round-3 K loop of the persistent GEMM (12 / 4 / 8 / 0 fragment reads per phase), for A/B through VLR_LIB
        build(force="--force" in sys.argv, defines=("VLR_KLOOP_BAL=0",), tag="_classic")
    else:
        build(force="--force" in sys.argv)
