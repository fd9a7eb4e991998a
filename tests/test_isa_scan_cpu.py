"""Build-time guard: no PACKED fp32 VALU instruction may consume an MFMA accumulator in place in the attention translation
units -- the one pattern that was necessary for the run-to-run differences of the fused dQ epilogue in round 4
(profiles/r05_nondeterminism.md).  Every csrc/*.hip is compiled to gfx950 assembly with the product's flags (hipcc cross-compiles on the
CPU box) and scanned by tools/pk_mfma_scan.py.  The GEMM translation units DO contain such instructions in their epilogues (bias add /
bf16 rounding of accumulator pairs); they have never differed run to run and every one of their outputs is under the step-level
bit-reproducibility tests (tests/parity_util.py::assert_step_bit_reproducible) -- their per-kernel site counts are pinned to a ceiling
here so that a compiler update that re-packs MORE accumulator arithmetic is noticed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CLEAN = ("qfx_attn.hip", "qfx_attn64.hip", "qfx_attn_bwd1.hip", "qfx_skinny.hip", "qfx_elem.hip", "qfx_cond.hip")
PINNED_MAX = {"qfx_gemm.hip": 64, "qfx_gemm_fp8.hip": 32}      # sites per kernel instantiation as of ROCm 7.2 / round 6


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    import __graft_entry__ as g
    out = tmp_path_factory.mktemp("isa")
    procs = {}
    for src in g.SOURCES:
        dst = os.path.join(out, src.replace(".hip", ".s"))
        cmd = [g._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + g.CSRC,
               *g.EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(g.CSRC, src), "-o", dst]
        procs[src] = (dst, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    res = {}
    for src, (dst, p) in procs.items():
        log, _ = p.communicate()
        assert p.returncode == 0, f"{src}:\n{log}"
        res[src] = dst
    return res


def test_every_source_is_scanned(asm):
    import __graft_entry__ as g
    assert set(asm) == set(g.SOURCES) == set(CLEAN) | set(PINNED_MAX)


def test_no_packed_fp32_valu_on_mfma_results_in_the_attention_and_skinny_kernels(asm):
    import pk_mfma_scan
    for src in CLEAN:
        sites = pk_mfma_scan.scan(asm[src])
        assert not sites, (src, {k: v[:4] for k, v in sites.items()})


def test_gemm_epilogue_site_counts_do_not_grow(asm):
    import pk_mfma_scan
    for src, cap in PINNED_MAX.items():
        sites = pk_mfma_scan.scan(asm[src])
        worst = max((len(v) for v in sites.values()), default=0)
        assert worst <= cap, (src, worst, cap)


def test_gemm_landing_buffer_epilogue_has_only_counted_vector_memory_operations(asm):
    """Round 6: the d(GELU) epilogue's landing-buffer side waits for its LDS-DMA requests by COUNT (s_waitcnt vmcnt(1 | 2 | 3): vector
    memory retires in order).  That is only right while the side's vector-memory instructions are exactly one request (D) and one store
    (S) per lane slot in that order -- a scratch reload or a plain load there shifts the count and its own compiler wait drains the
    request just issued (seen three times while writing it).  Pin the instruction order of both instantiations that have the buffer:
    MI 16-row passes x 2 lane slots of 'wait, request, store'."""
    import gemm_epi_vmem_seq as g
    text = open(asm["qfx_gemm.hip"]).read()
    for key, mi in (("dgelu 256x256", 8), ("dgelu 160x192", 5)):
        side = g.landing_side(g.seq(text, g.KERNELS[key]))
        want = ["w1", "D", "S", "w2", "D", "S"] + ["w3", "D", "S"] * (2 * mi - 2)
        # (the last slot's store may sit in another basic block -- hipcc lays the loop's exit path out elsewhere)
        assert side[:len(want) - 1] == want[:-1], (key, " ".join(side[:len(want) + 6]))
        assert side[len(want):len(want) + 1] != ["S"], (key, "a store beyond the last lane slot")
    # the gate + residual epilogue has the buffer only in A/B builds (QFX_GEMM_AUX_DMA bit 1, measured slower)
    for key in ("gate_res 256x256", "gate_res 160x192"):
        assert g.landing_side(g.seq(text, g.KERNELS[key])) == [], key
