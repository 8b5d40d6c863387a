"""CPU tier: host logic of the drop-in boundary and the C-ABI library's surface.
No compute call is made (there is no GPU here and the product has no CPU path)."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden
from primekg_rgcn_linkprediction_amd import (DrugDiseaseModel, DrugDiseaseRGCN, LinkPredictor, RGCNConv, _lib,
                                             ops, synth)


# ------------------------------------------------------------------ C ABI surface
def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rgcn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith(("rgcn_", "distmult_"))))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rgcn_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared, "ctypes prototypes and header disagree"
    assert lib.rgcn_abi_version() == _lib.ABI_VERSION


def test_error_strings():
    for code in (0, -1, -2, -3, -4, -5):
        assert _lib.strerror(code) and "unknown" not in _lib.strerror(code)
    with pytest.raises(IndexError):
        _lib.check(_lib.RGCN_ERR_RANGE, "x")
    with pytest.raises(ValueError):
        _lib.check(_lib.RGCN_ERR_ARG, "x")
    with pytest.raises(RuntimeError):
        _lib.check(_lib.RGCN_ERR_HIP, "x")


def test_null_handle_and_bad_args_return_codes_without_a_gpu():
    lib = _lib.load()
    assert lib.rgcn_graph_num_edges(None) == -1
    assert lib.rgcn_aggregate_workspace_bytes(None, 0, 64) == 0
    assert lib.rgcn_aggregate(None, 0, None, 64, None, None, 0, None) == _lib.RGCN_ERR_ARG
    assert lib.distmult_fwd(None, None, 1, None, None, 1, None, None, 1, 4, 6, None, None) == _lib.RGCN_ERR_ARG
    assert lib.distmult_fwd(None, None, 1, None, None, 1, None, None, 1, 0, 8, None, None) == _lib.RGCN_OK
    assert lib.rgcn_transform_bwd_params_workspace_bytes(30926, 3, 128, 128) > 0
    # every entry point rejects bad sizes / null pointers before it touches the device
    import ctypes
    out = ctypes.c_void_p()
    E, A, U = _lib.RGCN_ERR_ARG, _lib.RGCN_ERR_ARG, _lib.RGCN_ERR_UNSUPPORTED
    assert lib.rgcn_graph_create(None, None, -1, 5, 3, None, ctypes.byref(out)) == E
    assert lib.rgcn_graph_create(None, None, 4, 5, 3, None, ctypes.byref(out)) == E         # edges but no arrays
    assert lib.rgcn_graph_create(None, None, 0, 5, 0, None, ctypes.byref(out)) == E         # R = 0
    assert lib.rgcn_graph_create(None, None, 0, 5, 3, None, None) == E                      # no place for the handle
    assert lib.rgcn_graph_create(None, None, 0, 1 << 31, 3, None, ctypes.byref(out)) == U   # N*R over int32
    assert lib.rgcn_graph_create_bipartite(None, None, None, 2, 3, 3, 1, None, None, ctypes.byref(out)) == E
    assert lib.rgcn_graph_import(0, 3, 2, None, None, None, None, None, None, None, None, None, ctypes.byref(out)) == E
    assert lib.rgcn_graph_export(None, 0, None, None, None, None, None) == E
    assert out.value is None
    assert lib.rgcn_transform_fwd(None, None, None, None, None, 0, None, 10, 3, 6, 8, None, None) == A   # d_in % 4
    assert lib.rgcn_transform_fwd(None, None, None, None, None, 0, None, 10, 3, 8, 8, None, None) == A   # null operands
    assert lib.rgcn_transform_bwd_input(None, None, None, None, None, None, 10, 0, 8, 8, None, None) == A
    assert lib.rgcn_transform_bwd_params(None, None, None, None, 10, 3, 8, 8, None, None, None, None, 0, None) == A
    assert lib.distmult_bwd(None, None, None, 1, None, None, 1, None, None, 1, 4, 8, None, None, None, None, 0, 0, None) == E
    assert lib.distmult_bce_fwd(None, None, 1, None, None, 1, None, None, 1, None, 4, 8, None, None, None) == E
    assert lib.distmult_bce_bwd(None, None, None, None, None, 1, None, None, 1, None, None, 1, 4, 8, None, None, None, None, 0, 0, None) == E
    assert lib.distmult_bce_fwd(None, None, 1, None, None, 1, None, None, 1, None, 0, 8, None, None, None) == _lib.RGCN_OK
    assert lib.distmult_rank_tails(None, None, None, None, 4, 100, 48, None, None) != _lib.RGCN_OK       # d % 32
    assert lib.rgcn_sample_batch(None, None, 10, None, None, 4, 1, 0, None, None, None, None, None, None) == E
    assert lib.rgcn_sample_batch(None, None, 10, None, None, 4, 1, 100, None, None, None, None, None, None) == E
    assert lib.rgcn_sample_batch(None, None, 10, None, None, 0, 1, 100, None, None, None, None, None, None) == _lib.RGCN_OK


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librgcn_hip.so")
    with pytest.raises(_lib.RGCNLibraryError, match="no CPU"):
        _lib.load()
    assert not _lib.available()


# ------------------------------------------------------------------ RGCNConv contract (PyG signature)
def test_rgcnconv_parameters_and_init():
    torch.manual_seed(0)
    c = RGCNConv(64, 128, 3)
    assert [n for n, _ in c.named_parameters()] == ["weight", "root", "bias"]
    assert c.weight.shape == (3, 64, 128) and c.root.shape == (64, 128) and c.bias.shape == (128,)
    assert c.comp is None
    a = (6.0 / (64 + 128)) ** 0.5
    assert c.weight.abs().max() <= a and c.weight.abs().max() > 0.9 * a and c.root.abs().max() <= a
    assert torch.count_nonzero(c.bias) == 0
    b = RGCNConv(in_channels=64, out_channels=256, num_relations=3, num_bases=4)
    assert b.weight.shape == (4, 64, 256) and b.comp.shape == (3, 4)
    assert [n for n, _ in b.named_parameters()] == ["weight", "comp", "root", "bias"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the basis composition is a kernel too
        b.effective_weight()
    nb = RGCNConv(8, 8, 2, root_weight=False, bias=False)
    assert nb.root is None and nb.bias is None


def test_rgcnconv_same_init_stream_as_oracle():
    """same constructor order + same RNG draws as the PyG-equivalent oracle module"""
    from oracle import rgcn_oracle as O
    torch.manual_seed(7)
    mine = RGCNConv(16, 32, 3, num_bases=2)
    torch.manual_seed(7)
    ref = O.RGCNConvRef(16, 32, 3, num_bases=2)
    for (n1, p1), (n2, p2) in zip(mine.named_parameters(), ref.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_rgcnconv_errors():
    with pytest.raises(ValueError, match="both"):
        RGCNConv(4, 4, 3, num_bases=2, num_blocks=2)
    with pytest.raises(NotImplementedError):
        RGCNConv(4, 4, 3, num_blocks=2)
    with pytest.raises(NotImplementedError):
        RGCNConv(4, 4, 3, aggr="add")
    c = RGCNConv(4, 4, 3)
    ei, et = torch.zeros(2, 1, dtype=torch.long), torch.zeros(1, dtype=torch.long)
    with pytest.raises(AssertionError):
        c(torch.randn(2, 4), ei, None)
    with pytest.raises(ValueError):
        c(torch.randn(2, 8), ei, et)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        c(torch.randn(2, 4), ei, et)                  # CPU tensors: loud failure, no fallback
    with pytest.raises(TypeError):
        c(torch.zeros(2, 4, dtype=torch.long), ei, et)


# ------------------------------------------------------------------ model mirror (rgcn.py classes)
def test_model_state_dict_keys_and_param_count():
    m = DrugDiseaseModel(30926, 3)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 2078208
    assert list(m.state_dict()) == [
        "encoder.node_embeddings.weight", "encoder.conv1.weight", "encoder.conv1.root",
        "encoder.conv1.bias", "encoder.conv2.weight", "encoder.conv2.root", "encoder.conv2.bias",
        "decoder.relation_embeddings.weight"]
    assert isinstance(m.encoder, DrugDiseaseRGCN) and isinstance(m.decoder, LinkPredictor)
    assert m.encoder.dropout.p == 0.5 and m.decoder.dropout.p == 0.0


def test_reference_state_dict_loads():
    """state dict produced by the REFERENCE's DrugDiseaseModel (fixture) loads strictly."""
    z = load_golden("ref_model_eval.npz")
    sd = {k[4:].replace("__", "."): v for k, v in z.items() if k.startswith("sd__")}
    m = DrugDiseaseModel(100, 3, 64, 128)
    m.load_state_dict(sd, strict=True)
    zb = load_golden("ref_model_bases.npz")
    sdb = {k[4:].replace("__", "."): v for k, v in zb.items() if k.startswith("sd__")}
    mb = DrugDiseaseModel(60, 3, 64, 32, num_bases=4)
    mb.load_state_dict(sdb, strict=True)


def test_model_init_matches_reference_run_seed():
    """same RNG consumption order as the reference's constructor (embedding, conv1, conv2,
    embedding re-init, decoder): with the reference's seed the parameters are identical."""
    z = load_golden("ref_model_eval.npz")
    torch.manual_seed(4321)
    m = DrugDiseaseModel(num_nodes=100, num_relations=3, embedding_dim=64, hidden_dim=128,
                         dropout=0.5, decoder_dropout=0.0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, z["sd__" + k.replace(".", "__")]), k


def test_link_predictor_surface():
    d = LinkPredictor(3, 128, dropout=0.1)
    assert d.relation_embeddings.weight.shape == (3, 128)
    a = (6.0 / (3 + 128)) ** 0.5
    assert d.relation_embeddings.weight.abs().max() <= a
    h = torch.randn(4, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the [B, N] score matrix is a kernel too
        d.score_all_tails(h, torch.tensor([0, 1, 2, 0]), torch.randn(10, 128))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d(h, h, torch.tensor([0, 1, 2, 0]))


# ------------------------------------------------------------------ synthetic graphs
def test_primekg_like_shape():
    ei, et, n, r = synth.primekg_like(num_edges=20000, seed=42)
    assert (n, r) == (30926, 3) and ei.shape == (2, 20000) and ei.dtype == torch.int64
    assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0)) and torch.equal(et[0::2], et[1::2])
    lo_drug, lo_gene = synth.N_DISEASE, synth.N_DISEASE + synth.N_DRUG
    u, v, t = ei[0, 0::2], ei[1, 0::2], et[0::2]
    assert ((u[t == 0] >= lo_drug) & (u[t == 0] < lo_gene) & (v[t == 0] >= lo_gene)).all()
    assert ((u[t == 1] >= lo_gene) & (v[t == 1] < lo_drug)).all()
    assert ((u[t == 2] >= lo_gene) & (v[t == 2] >= lo_gene)).all()
    frac = torch.bincount(t, minlength=3).double() / t.numel()
    assert abs(frac[2] - 0.752) < 0.01 and abs(frac[1] - 0.188) < 0.01
    ei2, et2, *_ = synth.primekg_like(num_edges=20000, seed=42)
    assert torch.equal(ei, ei2) and torch.equal(et, et2)
    assert torch.bincount(ei[1]).max() > 50          # heavy tail


def test_bucket_input_validation_on_host():
    ei = torch.zeros(3, 4, dtype=torch.long)
    with pytest.raises(ValueError):
        ops.BucketedGraph(ei, torch.zeros(4, dtype=torch.long), 5, 2)
    with pytest.raises(ValueError):
        ops.BucketedGraph(torch.zeros(2, 4, dtype=torch.long), torch.zeros(3, dtype=torch.long), 5, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.BucketedGraph(torch.zeros(2, 4, dtype=torch.long), torch.zeros(4, dtype=torch.long), 5, 2)


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/rgcn_hip.h compiles as C99 and as C++ (-pedantic), and a C program linked against
    librgcn_hip.so sees the ABI version the header declares - the boundary is a real C ABI."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "include/rgcn_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %d %s\\n", rgcn_abi_version(), RGCN_ABI_VERSION, rgcn_strerror(-2));\n'
                   '  return rgcn_graph_num_edges(0) == -1 ? 0 : 1; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++17")):
        subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-fsyntax-only", "-x", "c" if cc == "gcc" else "c++",
                        "-I", root, str(src)], check=True)
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-I", root, str(src), "-o", str(exe), "-L", lib_dir, "-l:librgcn_hip.so",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,--allow-shlib-undefined"], check=True)
    import torch as _torch                                     # its bundled HIP runtime satisfies the .so at run time
    hip_dir = os.path.join(os.path.dirname(_torch.__file__), "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=hip_dir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env).stdout.split(maxsplit=2)
    assert out[0] == out[1] == str(_lib.ABI_VERSION) and "outside" in out[2]


class _NotATensor:                                     # something the restricted unpickler must refuse
    pass


def test_load_model_uses_the_restricted_unpickler(tmp_path):
    """checkpoints written by the trainer hold tensors, plain containers and ONE argparse.Namespace
    (train.py:431-442): `load_model` reads them with weights_only=True + that one allowed class; a file
    holding any other object is refused unless the caller opts in (no arbitrary unpickling)."""
    import argparse
    from primekg_rgcn_linkprediction_amd import DrugDiseaseModel, evaluate as E
    torch.manual_seed(0)
    model = DrugDiseaseModel(num_nodes=50, num_relations=3, embedding_dim=16, hidden_dim=32)
    args = argparse.Namespace(embedding_dim=16, hidden_dim=32, dropout=0.5, decoder_dropout=0.1, num_bases=None)
    good = {"epoch": 3, "model_state_dict": model.state_dict(), "optimizer_state_dict": {"state": {}, "param_groups": []},
            "best_val_loss": 0.25, "best_val_acc": 0.9, "train_losses": [0.7, 0.5], "args": args}
    torch.save(good, tmp_path / "good.pt")
    loaded, info = E.load_model(str(tmp_path / "good.pt"), torch.device("cpu"))
    assert info["epoch"] == 3 and info["num_nodes"] == 50 and info["best_val_loss"] == 0.25
    for k, v in model.state_dict().items():
        assert torch.equal(v, loaded.state_dict()[k])
    torch.save(dict(good, extra=_NotATensor()), tmp_path / "bad.pt")
    with pytest.raises(RuntimeError, match="restricted loader refuses"):
        E.load_model(str(tmp_path / "bad.pt"), torch.device("cpu"))
    loaded2, _ = E.load_model(str(tmp_path / "bad.pt"), torch.device("cpu"), trust_pickle=True)
    assert torch.equal(loaded2.state_dict()["encoder.conv1.weight"], model.state_dict()["encoder.conv1.weight"])


# ---------------------------------------------------------------------------------- build-time checks (no GPU)
def _csrc():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.path.join(root, "primekg_rgcn_linkprediction_amd", "csrc")


def test_no_kernel_reads_an_lds_fragment_before_its_wait():
    """The transform / fused-layer kernels cover inline-asm ``ds_read``s with hand-counted ``s_waitcnt lgkmcnt``;
    the compiler sees neither and once hoisted a copy of a fragment register above its wait (commit 7606cc0: wrong
    bits one run in five).  ``tools/check_waitcnt.py`` walks the gfx950 disassembly of EVERY kernel in the built
    objects (control flow followed, LDS returns in order, scalar loads out of order) and must find no instruction
    touching a register whose read is still uncovered - this is that bug class caught at build time."""
    import glob
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(_csrc()), "..", "tools"))
    import check_waitcnt as W
    objs = sorted(glob.glob(os.path.join(_csrc(), "build", "*.o")))
    assert len(objs) >= 9, "build the library first (conftest does)"
    kernels = reads = 0
    for path in objs:
        found = W.check_object(path)
        assert not found, {k: v[:3] for k, v in found.items()}
        k, r = W.stats(path)
        kernels, reads = kernels + k, reads + r
    assert kernels > 200 and reads > 3000                      # the walk really saw the kernels (incl. rocPRIM's)


def test_waitcnt_checker_catches_the_bug_class():
    """the checker on hand-made listings: the 7606cc0 shape (a copy of a fragment register above the covering
    wait), an uncovered use after a counted wait that is one too large, a hazard carried round a loop, and
    scalar loads, which make a non-zero count prove nothing; and the correct forms of each, which pass."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(_csrc()), "..", "tools"))
    import check_waitcnt as W

    def listing(body):
        lines = ["0000000000001000 <k>:"]
        for i, inst in enumerate(body):
            lines.append(f"\t{inst:<58} // {0x1000 + 4 * i:012X}: 00000000")
        return "\n".join(lines) + "\n"

    def violations(body):
        return W.check_function(W.parse_disassembly(listing(body))["k"])

    reads = ["ds_read_b128 v[2:5], v1", "ds_read_b128 v[6:9], v1 offset:256", "ds_read_b128 v[10:13], v1 offset:512"]
    ok = reads + ["s_waitcnt lgkmcnt(2)", "v_mov_b32_e32 v20, v2", "s_waitcnt lgkmcnt(0)",
                  "v_mfma_f32_32x32x16_f16 a[0:15], v[6:9], v[10:13], a[0:15]", "s_endpgm"]
    assert violations(ok) == []
    hoisted = reads + ["v_mov_b32_e32 v20, v6", "s_waitcnt lgkmcnt(0)", "s_endpgm"]          # the copy sits above the wait
    assert len(violations(hoisted)) == 1 and "v6" in violations(hoisted)[0]
    short = reads + ["s_waitcnt lgkmcnt(2)", "v_add_f32_e32 v21, v6, v6", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(violations(short)) == 1                                                        # lgkmcnt(2) covers the first read only
    # loop: the read issued at the bottom is used at the top of the next trip; `s_cbranch_scc1 <k+0x4>` jumps back
    loop = ["s_waitcnt lgkmcnt(0)", "v_add_f32_e32 v30, v2, v2", "ds_read_b128 v[2:5], v1",
            "s_cbranch_scc1 65533                                   // 00000000100C: 00000000 <k+0x4>", "s_endpgm"]
    text = listing(loop[:3]) + f"\t{loop[3]}\n\ts_endpgm                                    // 000000001010: 00000000\n"
    assert len(W.check_function(W.parse_disassembly(text)["k"])) == 1
    fixed = listing(["v_add_f32_e32 v30, v30, v30", "s_waitcnt lgkmcnt(0)", "v_add_f32_e32 v30, v2, v2"]) \
        + "\tds_read_b128 v[2:5], v1                                    // 00000000100C: 00000000\n" \
        + "\ts_cbranch_scc1 65532                                       // 000000001010: 00000000 <k+0x4>\n" \
        + "\ts_endpgm                                                   // 000000001014: 00000000\n"
    assert W.check_function(W.parse_disassembly(fixed)["k"]) == []
    smem = ["ds_read_b32 v2, v1", "s_load_dword s4, s[0:1], 0x0", "ds_read_b32 v3, v1", "s_waitcnt lgkmcnt(1)",
            "v_add_f32_e32 v9, v2, v2", "s_endpgm"]
    assert len(violations(smem)) == 1                                                         # out-of-order scalar return
    assert violations(smem[:3] + ["s_waitcnt lgkmcnt(0)"] + smem[4:]) == []
    # vector-memory loads under a counted vmcnt (round 4: operand maxima requested ahead of the LDS-DMAs of a GEMM prologue):
    # four DMAs behind two loads - vmcnt(4) covers both loads, a copy above the wait or a DMA skipped on one path does not
    loads = ["global_load_dword v4, v[2:3], off", "global_load_dword v5, v[2:3], off offset:256"]
    dmas = ["global_load_lds_dwordx4 v[8:9], off"] * 4
    assert violations(loads + dmas + ["s_waitcnt vmcnt(4)", "v_max_f32_e32 v6, v4, v5", "s_endpgm"]) == []
    copied = loads + dmas + ["v_mov_b32_e32 v20, v5", "s_waitcnt vmcnt(4)", "v_max_f32_e32 v6, v4, v20", "s_endpgm"]
    assert len(violations(copied)) == 1 and "v5" in violations(copied)[0]
    assert len(violations(loads + dmas + ["s_waitcnt vmcnt(5)", "v_max_f32_e32 v6, v4, v5", "s_endpgm"])) == 1   # v5 may be in flight
    # one DMA sits behind a branch that can skip it: on that path only three are younger than the loads
    skipped = listing(loads + dmas[:3]) \
        + "\ts_cbranch_execz 1                                          // 000000001014: 00000000 <k+0x1c>\n" \
        + "\tglobal_load_lds_dwordx4 v[8:9], off                        // 000000001018: 00000000\n" \
        + "\ts_waitcnt vmcnt(4)                                         // 00000000101C: 00000000\n" \
        + "\tv_max_f32_e32 v6, v4, v5                                   // 000000001020: 00000000\n" \
        + "\ts_endpgm                                                   // 000000001024: 00000000\n"
    assert len(W.check_function(W.parse_disassembly(skipped)["k"])) == 1


def test_editing_any_header_rebuilds_the_objects_that_include_it(tmp_path):
    """csrc/Makefile takes its header prerequisites from the compiler (-MMD): touching rgcn_hub_finish.h - the one
    definition that makes k_reduce_partials and the transform prologue "the same bits" - must put exactly the two
    objects that include it back on the build list (round 2's hand-written list had left it out)."""
    import os
    import subprocess
    csrc = _csrc()
    header = os.path.join(csrc, "rgcn_hub_finish.h")
    st = os.stat(header)
    try:
        subprocess.run(["make", "-C", csrc, "-q"], check=False)
        os.utime(header)                                          # "edited now"
        plan = subprocess.run(["make", "-C", csrc, "-n"], check=True, capture_output=True, text=True).stdout
    finally:
        os.utime(header, (st.st_atime, st.st_mtime))
    rebuilt = {w[len("build/"):-2] for line in plan.splitlines() if " -c " in line for w in line.split() if w.startswith("build/") and w.endswith(".o")}
    assert rebuilt == {"rgcn_aggregate", "rgcn_transform_split"}, rebuilt
