"""The C-ABI library loads and exports every symbol include/occ_hip.h declares (no compute calls)."""
import ctypes
import os

import pytest

from occm_amd import _lib


def test_header_parses_and_has_the_core_entry_points():
    protos = _lib.parse_header()
    for name in ("occ_last_error", "occ_version", "occ_arch", "occ_rawboost_fir_bank", "occ_rawboost_center_norm",
                 "occ_compactness_loss", "occ_ce_loss", "occ_adam_multi", "occ_gemm", "occ_layernorm"):
        assert name in protos, name
    assert protos["occ_last_error"][0] is ctypes.c_char_p
    assert len(protos["occ_rawboost_fir_bank"][1]) == 11


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    for name in _lib.parse_header():
        assert hasattr(handle, name), "libocc_hip.so does not export %s" % name
    assert handle.occ_version() >= 100
    assert handle.occ_arch() == b"gfx950"


def test_host_argument_validation_needs_no_gpu():
    handle = _lib.lib()
    rc = handle.occ_rawboost_fir_bank(None, 0, None, None, None, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b"null pointer" in handle.occ_last_error()
    with pytest.raises(_lib.OccError):
        _lib.check(rc, "occ_rawboost_fir_bank")


# kernels on the hot path: a register spill in one of these costs tens of percent (a 20-byte spill in the default GEMM took the bench
# from 10.2 to 12.1 ms before it was noticed), so the build's resource report is checked instead of waiting for a timing run
_NO_SCRATCH = ["gemm_bf16_dma_kernelILi128ELb0ELi128", "gemm_bf16_dma_kernelILi128ELb1ELi128", "gemm_bf16_dma_kernelILi256", "gemm_bf16_hs_kernel", "gemm_bf16_ks2_kernel", "gemm_bf16_dma_n64_kernel",
               "gemm_kernelILi2ELi64", "gemm_kernelILi2ELi128", "gemm_tn_bf16_kernel", "gemm_tn_dma_kernelILi4", "gemm_tn_dma_kernelILi1", "attention_mfma_head_kernelILi7ELi8", "attention_mfma_head_kernelILi7ELi4",
               "attention_mfma_long_kernelILi64", "attention_mfma_kernelILi4", "conv0_ln_gelu_kernelItLi10", "layernorm_kernelIftLi2", "conv0_bwd_kernel",
               "fir_bank_kernel", "adam_multi_kernel",
               # round 3: this round's hot kernels (the eight-phase GEMMs in all formats, the weight-gradient kernels, attention backward incl.
               # its dropout form, the fused LayerNorm backward of the transformer layers and the conv stack's LayerNorm+GELU backward)
               "gemm_p8_kernel", "gemm_tn_p8_kernel", "gemm_tn_p8_pair_kernel", "attention_bwd2_kernelILi64ELb0", "attention_bwd2_kernelILi80ELb0", "attention_bwd2_kernelILi64ELb1", "layernorm_bwd16_kernelItfLi2ELb0ELi12ELb1",
               "layernorm_bwd16_kernelItfLi3ELb0ELi8ELb1", "layernorm_bwd16_kernelIttLi1ELb1", "layernorm_kernelIftLi2ELb1", "attention_mfma_long_kernel",
               # round 4: the separate-pass LayerNorm backward at C = 1024 (the route residual dropout / layerdrop take: 12 waves, the 16-wave
               # form spilled) and the four-wave GEMM
               "Li2ELb0ELi12ELb0EEEv", "gemm_q4_kernel", "conv0_mfma_kernel", "conv0_bwd_mfma_kernel", "gemm_kernelILi4ELi128"]


def test_hot_kernels_do_not_spill():
    build_dir = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "_build")
    reports = [f for f in (os.listdir(build_dir) if os.path.isdir(build_dir) else []) if f.endswith(".resources.txt")]
    if not reports:                                   # library built by an older Makefile (or shipped prebuilt): rebuild to get the reports
        import subprocess
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc"), "-B", "-j4"], check=True, capture_output=True)
        reports = [f for f in os.listdir(build_dir) if f.endswith(".resources.txt")]
    kernels, name = {}, None
    for f in reports:
        for line in open(os.path.join(build_dir, f)):
            key, _, val = line.strip().partition(": ")
            if key == "Function Name":
                name = val
                kernels[name] = {}
            elif name:
                kernels[name][key.split(" [")[0]] = val
    assert len(kernels) > 100
    for want in _NO_SCRATCH:
        hits = [k for k in kernels if want in k]
        assert hits, "no kernel matching %s in the build report" % want
        for k in hits:
            assert kernels[k]["ScratchSize"] == "0", "%s spills %s bytes per lane" % (k, kernels[k]["ScratchSize"])


def test_ctypes_struct_mirrors_match_header(tmp_path):
    """Every ctypes.Structure in occm_amd._lib mirrors a struct of include/occ_hip.h: same size and same field offsets as gcc lays them out
    (a field added to the header but not to the mirror would silently shift everything behind it)."""
    import ctypes
    import re
    import subprocess
    from occm_amd import _lib
    pairs = {"occ_rowmap": _lib.RowMap, "occ_gemm_desc": _lib.GemmDesc, "occ_finalize_job": _lib.FinalizeJob, "occ_gemm_tn_desc": _lib.GemmTnDesc,
             "occ_master_desc": _lib.MasterDesc, "occ_master_grads": _lib.MasterGrads, "occ_readout_desc": _lib.ReadoutDesc, "occ_readout_grads": _lib.ReadoutGrads}
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "occ_hip.h")
    text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "%s"' % hdr, "int main(void) {"]
    for cname, cls in pairs.items():
        body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (cname, cname), text, flags=re.S)
        assert body, cname
        names = []
        for decl in body.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            first = re.search(r"(\w+)\s*(\[\w*\])?\s*$", parts[0]).group(1)
            names += [first] + [re.search(r"(\w+)\s*(\[\w*\])?\s*$", p).group(1) for p in parts[1:]]
        assert names == [f[0] for f in cls._fields_], (cname, names, [f[0] for f in cls._fields_])
        lines.append('  printf("%s %%zu", sizeof(%s));' % (cname, cname))
        lines += ['  printf(" %%zu", offsetof(%s, %s));' % (cname, n) for n in names]
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in out:
        tok = line.split()
        cls = pairs[tok[0]]
        assert int(tok[1]) == ctypes.sizeof(cls), tok[0]
        assert [int(t) for t in tok[2:]] == [getattr(cls, f[0]).offset for f in cls._fields_], tok[0]
