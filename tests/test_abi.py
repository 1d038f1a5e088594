"""CPU: the C-ABI library loads and exports every symbol include/hypel.h declares; the ctypes signature
table of hypelcnn_amd/backend.py matches the header's parameter counts (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hypel.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|uint32_t|const char\*)\s+(hypel_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(2)] = n
    return out


@pytest.fixture(scope="module")
def lib():
    from hypelcnn_amd import backend
    if not os.path.exists(backend.LIB_PATH):
        backend.build_library()
    return backend.load_library()


def test_every_declared_symbol_is_exported(lib):
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in hypel.h but not exported"
    header = int(re.search(r"#define\s+HYPEL_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    from hypelcnn_amd import backend
    assert lib.hypel_version() == header == backend.ABI_VERSION
    assert lib.hypel_last_error() is not None


def test_signature_table_matches_header():
    from hypelcnn_amd import backend
    decl = _declared()
    for short, sig in backend.SIGNATURES.items():
        name = "hypel_" + short
        if name not in decl:
            continue
        assert decl[name] == len(sig) + 1, f"{name}: header has {decl[name]} params, binding {len(sig)} + stream"
    bound = {"hypel_" + k for k in backend.SIGNATURES}
    special = {"hypel_crc32c", "hypel_version", "hypel_last_error", "hypel_device_info", "hypel_gan_generator_blocks",
               "hypel_dense_stack_blocks", "hypel_gan_generator_blocks_apps", "hypel_dense_stack_blocks_apps",
               "hypel_dense_stack_supported", "hypel_gan_generator_tap_supported",
               "hypel_graph_begin_capture",
               "hypel_graph_end_capture", "hypel_graph_launch", "hypel_graph_destroy"}
    missing = set(decl) - bound - special
    assert not missing, f"header functions without a Python binding: {missing}"


def test_table_struct_layouts_match_header():
    from hypelcnn_amd.backend import GROUP_DTYPE, LOSS_TERM_DTYPE, SEG_DTYPE, TILE_DTYPE
    assert LOSS_TERM_DTYPE.itemsize == 104 and LOSS_TERM_DTYPE.fields["mode"][1] == 72 and LOSS_TERM_DTYPE.fields["slot"][1] == 100
    assert SEG_DTYPE.itemsize == 24 and GROUP_DTYPE.itemsize == 24 and TILE_DTYPE.itemsize == 56
    assert TILE_DTYPE.fields["n"][1] == 52  # hypel_tile_t.n (ABI 4) sits where `reserved` was
    from hypelcnn_amd.backend import COPY_BLOCK_DTYPE
    assert COPY_BLOCK_DTYPE.itemsize == 40 and COPY_BLOCK_DTYPE.fields["rows"][1] == 16
    assert SEG_DTYPE.fields["b_off"][1] == 8 and SEG_DTYPE.fields["k"][1] == 16
    assert GROUP_DTYPE.fields["seg_begin"][1] == 8 and GROUP_DTYPE.fields["rows"][1] == 16


def test_no_cpu_fallback_without_gpu():
    import torch
    from hypelcnn_amd.backend import HipBackend, HypelError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(HypelError):
        HipBackend()


def test_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "hypelcnn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", txt, flags=re.M), os.path.join(dp, f)
