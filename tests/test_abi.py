"""The C-ABI library loads on a CPU-only box and exports exactly what include/dualdiff_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dualdiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    fns = header_functions()
    for name in ("dd_gemm", "dd_attention", "dd_groupnorm_nhwc", "dd_layernorm", "dd_cfg_ddim_step",
                 "dd_timestep_embedding", "dd_add", "dd_conv3x3_small_cout"):
        assert name in fns


def test_library_exports_every_declared_symbol():
    from dualdiff_amd import _build, _native
    if not os.path.exists(_build.lib_path()):
        pytest.skip("library not built (run __graft_entry__.build())")
    lib = _native.load(build_if_missing=False)
    declared = header_functions()
    for name in declared:
        assert hasattr(lib, name), "%s declared in the header but not exported" % name
    assert sorted(_native.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    assert lib.dd_abi_version() == _native.ABI_VERSION == 4
    import ctypes
    for which, st in enumerate((_native.GemmDesc, _native.AttnDesc, _native.XAttnDesc, _native.Gemm8Desc,
                              _native.BoxTokensDesc)):
        assert lib.dd_desc_size(which) == ctypes.sizeof(st), st.__name__
    assert lib.dd_desc_size(99) == -1
    assert lib.dd_target_arch() == b"gfx950"
    assert b"workspace" in lib.dd_error_string(-4)


def test_validation_without_a_gpu():
    """Argument validation runs before any launch: bad descriptors are rejected on a CPU-only box."""
    import ctypes
    from dualdiff_amd import _build, _native
    if not os.path.exists(_build.lib_path()):
        pytest.skip("library not built")
    lib = _native.load(build_if_missing=False)
    d = _native.GemmDesc()
    assert lib.dd_gemm(ctypes.byref(d), None) == -1            # null pointers
    d.a = d.w = d.out = 16
    d.rows, d.n, d.k, d.k1, d.lda, d.ldc, d.alpha, d.dtype = 8, 12, 64, 64, 64, 12, 1.0, 1
    assert lib.dd_gemm(ctypes.byref(d), None) == -1            # n % 8 != 0
    a = _native.AttnDesc()
    a.q = a.k = a.v = a.o = 16
    a.batch, a.heads, a.head_dim, a.lq, a.lk = 1, 8, 64, 4, 4
    a.ldq = a.ldk = a.ldv = a.ldo = 512
    assert lib.dd_attention(ctypes.byref(a), None) == -2       # head_dim 64 unsupported


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from dualdiff_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.add(torch.zeros(8, dtype=torch.float16), torch.zeros(8, dtype=torch.float16))


def test_header_is_plain_c():
    """The boundary is a C ABI: include/dualdiff_hip.h must compile as C99 and as C++17 on its own (no torch / HIP types
    in the signatures, dd_stream_t is a void pointer)."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "dualdiff_hip.h")
    for cc, flags in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not installed")
        r = subprocess.run([cc] + flags + ["-fsyntax-only", "-Wall", "-Werror", hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    import re
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)                          # declarations without comments
    assert "torch" not in code.lower() and "at::" not in code and "hipStream_t" not in code
    assert "typedef void* dd_stream_t" in code and "#include <stdint.h>" in code


def test_c_host_links_and_validates(tmp_path):
    """A C host (what INTEGRATION.md §4 shows) compiled with gcc against include/dualdiff_hip.h and the built library:
    version / descriptor-size handshake and argument validation run without a GPU (no launch happens before them)."""
    import shutil
    import subprocess
    from dualdiff_amd import _build
    if shutil.which("gcc") is None or not os.path.exists(_build.lib_path()):
        pytest.skip("gcc or the built library missing")
    src = tmp_path / "host.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "dualdiff_hip.h"
int main(void) {
  dd_gemm_desc d; memset(&d, 0, sizeof d);
  if (dd_abi_version() != DD_ABI_VERSION) return 10;
  if (dd_desc_size(0) != (int64_t)sizeof(dd_gemm_desc) || dd_desc_size(1) != (int64_t)sizeof(dd_attn_desc)) return 11;
  if (dd_desc_size(4) != (int64_t)sizeof(dd_box_tokens_desc) || dd_desc_size(99) != -1) return 12;
  if (dd_gemm(&d, 0) != DD_ERR_BAD_ARG) return 13;                 /* null pointers: rejected before any launch */
  if (strcmp(dd_target_arch(), "gfx950") != 0) return 14;
  printf("%s\n", dd_error_string(DD_ERR_UNSUPPORTED));
  return 0;
}
""")
    exe = tmp_path / "host"
    libdir = os.path.dirname(_build.lib_path())
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-ldualdiff_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert "unsupported" in r.stdout
