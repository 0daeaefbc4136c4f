"""CPU: the C-ABI library loads and exports every symbol include/zuko_amd.h declares; host-side
logic (module trees, state_dict keys, masks, loud failure on CPU tensors).  No kernel is launched."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, T, golden


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zuko_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(zk_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import zuko_amd._C as C

    lib = ctypes.CDLL(C.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/zuko_amd.h but not exported"
    # and the Python binding table covers exactly the declared surface
    assert sorted(C.SIGNATURES) == syms


def test_import_fails_loudly_without_library(monkeypatch):
    import zuko_amd._C as C

    with pytest.raises(ImportError, match="no CPU"):
        C._Lib("/nonexistent/libzuko_amd.so")


def test_cpu_tensors_are_rejected():
    import zuko_amd.transforms as ZT
    from zuko_amd import ops

    t = ZT.MonotonicRQSTransform(torch.randn(8), torch.randn(8), torch.randn(7))
    with pytest.raises(RuntimeError, match="HIP device"):
        t(torch.randn(4))
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.linear(torch.randn(2, 3), torch.randn(4, 3))


def test_masks_and_module_tree():
    import zuko_amd.flows as F
    from zuko_amd.nn import MaskedLinear, masked_mlp_masks

    g = golden("masks.npz")
    t = F.MaskedAutoregressiveTransform(64, 0, shapes=[(8,), (8,), (7,)], hidden_features=(256, 256, 256))
    layers = [m for m in t.hyper if isinstance(m, MaskedLinear)]
    assert [tuple(m.weight.shape) for m in layers] == [(256, 64), (256, 256), (256, 256), (1472, 256)]
    for i, m in enumerate(layers):
        shape = tuple(g[f"ar64_shape{i}"])
        ref = np.unpackbits(g[f"ar64_mask{i}"])[: shape[0] * shape[1]].reshape(shape).astype(bool)
        assert np.array_equal(m.mask.numpy(), ref)
    assert int(sum(m.mask.sum() for m in layers)) == 265784  # nnz of the cfg2 conditioner
    keys = set(t.state_dict())
    assert {"order", "hyper.0.weight", "hyper.0.bias", "hyper.0.mask", "hyper.6.mask"} <= keys
    A = T(g["free_adjacency"])
    for i, m in enumerate(masked_mlp_masks(A, (16, 32))):
        assert np.array_equal(m.numpy(), g[f"free_mask{i}"])
    with pytest.raises(ValueError, match="null Jacobian"):
        masked_mlp_masks(torch.zeros(3, 3, dtype=bool), (8,))


def test_constructor_assertions_match_reference_messages():
    import zuko_amd.flows as F

    with pytest.raises(AssertionError, match="'order' should have 3 elements"):
        F.MaskedAutoregressiveTransform(3, order=[0, 1])
    with pytest.raises(AssertionError, match="ones on the diagonal"):
        F.MaskedAutoregressiveTransform(3, adjacency=torch.zeros(3, 3, dtype=bool))
    with pytest.raises(AssertionError, match="cycles"):
        F.MaskedAutoregressiveTransform(3, adjacency=torch.ones(3, 3, dtype=bool))
    t = F.MaskedAutoregressiveTransform(4, adjacency=torch.tril(torch.ones(4, 4, dtype=bool)))
    assert t.passes == 4
    t = F.MaskedAutoregressiveTransform(6, passes=2)
    assert t.passes == 2 and t.order.tolist() == [0, 0, 0, 1, 1, 1]
    assert isinstance(F.MaskedAutoregressiveTransform(1, 3), F.ElementWiseTransform)


def test_flow_state_dict_layout_and_pickle(tmp_path):
    import zuko_amd.flows as F

    torch.manual_seed(0)
    flow = F.NSF(3, 5, transforms=2, hidden_features=[16, 16])
    keys = list(flow.state_dict())
    assert "transform.transforms.0.order" in keys and "transform.transforms.1.hyper.4.mask" in keys
    assert "base.loc" in keys and "base.scale" in keys
    p = tmp_path / "flow.pt"
    torch.save(flow, p)  # whole-module pickle, as tests/test_flows.py:78-91 of the reference does
    again = torch.load(p, weights_only=False)
    for (k1, v1), (k2, v2) in zip(flow.state_dict().items(), again.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert "MaskedAutoregressiveTransform" in repr(flow)


def test_product_code_never_touches_the_oracle_or_a_cpu_fallback():
    """The oracle is test infrastructure: nothing under zuko_amd/ may import it, and the product path must fail
    loudly (not fall back) on CPU tensors."""
    import pathlib
    import re

    root = pathlib.Path(__file__).resolve().parents[1]
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|zuko_oracle", re.M)
    for f in (root / "zuko_amd").rglob("*.py"):
        assert not pat.search(f.read_text()), f"{f} references the oracle"
    import torch

    from zuko_amd import ops

    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.rqs_forward(torch.zeros(4, 2), torch.zeros(4, 2, 8), torch.zeros(4, 2, 8), torch.zeros(4, 2, 7))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))


def test_ctypes_argument_blocks_match_the_c_compiler(tmp_path):
    """zuko_amd/_C.py builds its ctypes.Structure classes by PARSING include/zuko_amd.h; their sizes and every field offset must be what a
    C compiler makes of the same header (gcc: the header is plain C), for the versioned argument blocks and the descriptor arrays."""
    import ctypes
    import subprocess

    import zuko_amd._C as C

    names = sorted(C.STRUCTS)
    assert {"zk_ar_args_v1", "zk_coupling_args_v1", "zk_ar_inc_args_v1", "zk_wgrad_layer_v1", "zk_gather_desc_v1"} <= set(names)
    header = os.path.join(ROOT, "include", "zuko_amd.h")
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{header}"', "int main(void) {"]
    for n in names:
        lines.append(f'  printf("{n} %zu\\n", sizeof({n}));')
        for f, _ in C.STRUCTS[n]._fields_:
            lines.append(f'  printf("{n}.{f} %zu\\n", offsetof({n}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "sizes.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n in names:
        cls = C.STRUCTS[n]
        assert int(out[n]) == ctypes.sizeof(cls), n
        for f, _ in cls._fields_:
            assert int(out[f"{n}.{f}"]) == getattr(cls, f).offset, f"{n}.{f}"


def test_entry_points_reject_foreign_argument_blocks_without_touching_the_device():
    """A block with the wrong struct_size / version (a caller built against another header) or an empty descriptor list is refused with
    hipErrorInvalidValue before anything is launched — this runs on a box without a GPU."""
    import zuko_amd._C as C

    lib = C.lib()
    EINVAL = 1
    for struct, fns in (("zk_ar_args_v1", ["zk_ar_forward", "zk_ar_forward_split", "zk_ar_forward_static", "zk_ar_forward_train", "zk_ar_inverse_sweep", "zk_ar_dgrad_chain", "zk_ar_dgrad_full", "zk_ar_backward_full"]),
                        ("zk_coupling_args_v1", ["zk_coupling_forward", "zk_coupling_inverse"]), ("zk_ar_inc_args_v1", ["zk_ar_inverse_incremental"])):
        good = C.args(struct)
        for fn in fns:
            bad = C.args(struct)
            if struct == "zk_ar_args_v1":
                # version 1 grew by a tail (phi_packed .. eps): the EARLIER size is still a valid block (missing fields read as zero) and must
                # take the same path as the full one; a block cut inside the original fields or longer than the library's is foreign
                first = C.STRUCTS[struct].phi_packed.offset
                short = C.args(struct)
                short.struct_size = first
                assert getattr(lib, fn)(short, None) == getattr(lib, fn)(C.args(struct), None), fn
                bad.struct_size = first - 8
                assert getattr(lib, fn)(bad, None) == EINVAL, fn
                bad.struct_size = good.struct_size + 8
            else:
                bad.struct_size = good.struct_size - 8
            assert getattr(lib, fn)(bad, None) == EINVAL, fn
            bad = C.args(struct)
            bad.version = 99
            assert getattr(lib, fn)(bad, None) == EINVAL, fn
    assert lib.zk_gather_multi(0, None, None) == EINVAL and lib.zk_gather_multi(9, None, None) == EINVAL
    assert lib.zk_wgrad_multi(0, None, 16, None) == EINVAL and lib.zk_wgrad_multi(5, None, 16, None) == EINVAL
    layer = (C.STRUCTS["zk_wgrad_layer_v1"] * 1)()
    layer[0].struct_size = 4
    assert lib.zk_wgrad_multi(1, ctypes.cast(layer, ctypes.c_void_p), 16, None) == EINVAL
    desc = (C.STRUCTS["zk_gather_desc_v1"] * 1)()
    desc[0].struct_size = 4
    assert lib.zk_gather_multi(1, ctypes.cast(desc, ctypes.c_void_p), None) == EINVAL


def test_jit_build_failures_never_raise_into_the_callers_log_prob(tmp_path, monkeypatch):
    """the first-use compile of a static-shape kernel writes .hip / .so / .json / lock files.  A directory that cannot be
    written (read-only install, full disk) must make `_build_so` return None — the caller then stays on the generic kernel — not
    raise OSError out of the user's call; ZUKO_AMD_CACHE_DIR redirects the JIT's output, the prebuilt kernels stay visible."""
    from zuko_amd import static_ar

    blocker = tmp_path / "not_a_directory"
    blocker.write_text("x")
    monkeypatch.setenv("ZUKO_AMD_CACHE_DIR", str(blocker))  # <file>/ars cannot be created: NotADirectoryError (an OSError)
    assert static_ar._jit_dir() == os.path.join(str(blocker), "ars") and static_ar._dirs()[-1] == static_ar.ARS_DIR
    assert static_ar._build_so("ars_unit_test_never_built", lambda: "int main() { return 0; }", {"so": "x"}, verbose=False) is None
    # the scan still finds what was built ahead of time next to the package
    static_ar._INDEX = None
    idx = static_ar._scan()
    assert idx and all(os.path.exists(os.path.join(m["dir"], m["so"])) for metas in idx.values() for m in metas)
    static_ar._INDEX = None
    # a writable cache directory takes the JIT's files (hipcc is present in the build container; skip the compile itself when it is not)
    good = tmp_path / "cache"
    monkeypatch.setenv("ZUKO_AMD_CACHE_DIR", str(good))
    assert static_ar._jit_dir() == os.path.join(str(good), "ars")
    assert static_ar._arch() == "gfx950"


def test_jit_threshold_counts_rows_over_calls(monkeypatch):
    """A conditioner without a kernel on disk gets one compiled when ONE batch reaches ZUKO_AMD_JIT_MIN_ROWS — or when the batches seen so far add up to
    eight times that (zuko_amd/static_ar.py: effective_rows): a training loop with small batches does not stay on the generic kernel for ever."""
    from zuko_amd import static_ar

    monkeypatch.setenv("ZUKO_AMD_JIT_MIN_ROWS", "1000")

    class Holder:
        pass

    h = Holder()
    assert static_ar.effective_rows(h, 5000) == 5000  # a large batch: itself
    h = Holder()
    got = [static_ar.effective_rows(h, 300) for _ in range(30)]
    first = next(i for i, g in enumerate(got) if g >= 1000)
    assert got[:first] == [300] * first and first == -(-8 * 1000 // 300) - 1 and all(g == 1000 for g in got[first:])

    class Slotted:
        __slots__ = ()

    assert static_ar.effective_rows(Slotted(), 300) == 300  # (nowhere to keep the count: the per-call rule)
