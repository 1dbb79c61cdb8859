"""The C-ABI library loads and exports every symbol include/avlen_hip.h declares, with the arity the
ctypes layer assumes.  No compute calls (CPU only)."""
import os
import re
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "avlen_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\n(?:int|size_t|void|const char\*)\s+(avlen_\w+)\s*\(([^;]*?)\)\s*;", src):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",")])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    from avlen_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 35
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(raw, name), f"missing export {name}"
        assert name in _lib.SIGNATURES, f"no ctypes signature for {name}"
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, len(_lib.SIGNATURES[name][1]), nargs)
    assert set(_lib.SIGNATURES) == set(decls)
    assert b"gfx950" in _lib.lib.avlen_build_info()


def test_struct_sizes_match_header_layout():
    from avlen_amd import _lib
    # avlen_linear: 2 pointers, 2 ints, pointer, int (+pad), pointer; avlen_conv: 2 pointers, 6 ints, pointer, int (+pad), 4 pointers
    assert ctypes.sizeof(_lib.Linear) == 48 and ctypes.sizeof(_lib.Conv) == 88 and ctypes.sizeof(_lib.Affine) == 16
    assert ctypes.sizeof(_lib.Mha) == 96


def test_struct_sizes_against_c_compiler(tmp_path):
    """sizeof() of every parameter-view struct as gcc sees the header == ctypes' layout."""
    import subprocess
    from avlen_amd import _lib
    names = {"avlen_linear": _lib.Linear, "avlen_conv": _lib.Conv, "avlen_affine": _lib.Affine,
             "avlen_resblock": _lib.ResBlock, "avlen_resnet18": _lib.ResNet18, "avlen_cnn3": _lib.Cnn3,
             "avlen_mha": _lib.Mha, "avlen_enc_layer": _lib.EncLayer, "avlen_dec_layer": _lib.DecLayer,
             "avlen_transformer": _lib.Transformer, "avlen_smt": _lib.Smt, "avlen_dialog": _lib.Dialog,
             "avlen_clip_block": _lib.ClipBlock, "avlen_clip_text": _lib.ClipText, "avlen_gru": _lib.Gru,
             "avlen_heads": _lib.Heads, "avlen_ln_fold": _lib.LnFold, "avlen_extmem_op": _lib.ExtMemOp,
             "avlen_cmd": _lib.Cmd}
    src = tmp_path / "s.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in names)
    src.write_text(f'#include <stdio.h>\n#include "avlen_hip.h"\nint main(void) {{\n{body}\n  return 0; }}\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n, cls in names.items():
        assert sizes[n] == ctypes.sizeof(cls), (n, sizes[n], ctypes.sizeof(cls))
