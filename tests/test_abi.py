"""The C ABI: every function declared in include/*.h is exported by libnerf_amd.so and bound in
nerf_amd/_lib.py; struct layouts in the ctypes mirror match the C headers.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
HEADERS = ["nerf_amd.h", "nerf_data.h", "nerf_mlp.h", "nerf_kernels.h"]


def declared_functions(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)      # comments
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*?(?<!\\)$", " ", text, flags=re.M | re.S)  # preprocessor lines
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    text = re.sub(r"typedef\s+struct\s*\w*\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    names = []
    for stmt in text.split(";"):
        m = re.search(r"\b([A-Za-z_]\w*)\s*\(", stmt)
        if m and "typedef" not in stmt and "{" not in stmt.split("(")[0]:
            names.append(m.group(1))
    return names


def exported_symbols():
    so = os.path.join(ROOT, "nerf_amd", "libnerf_amd.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


@pytest.mark.parametrize("header", HEADERS)
def test_every_declared_function_is_exported(header):
    names = declared_functions(os.path.join(INCLUDE, header))
    assert len(names) >= 5, (header, names)
    exported = exported_symbols()
    missing = [n for n in names if n not in exported]
    assert not missing, f"{header}: declared but not exported: {missing}"


def test_ctypes_binding_covers_every_declaration():
    from nerf_amd._lib import EXPORTED
    declared = set()
    for h in HEADERS:
        declared.update(declared_functions(os.path.join(INCLUDE, h)))
    unbound = sorted(declared - set(EXPORTED))
    assert not unbound, f"declared in include/ but without a ctypes signature: {unbound}"
    stale = sorted(set(EXPORTED) - declared)
    assert not stale, f"bound in _lib.py but not declared in include/: {stale}"


def test_library_loads_and_reports_abi_version():
    from nerf_amd._lib import lib
    assert lib.nerf_amd_abi_version() >= 1


def test_struct_layouts_match_the_headers(tmp_path):
    """Compile a C probe against include/*.h and compare sizeof/offsetof with the ctypes mirrors."""
    from nerf_amd._lib import MLP, Camera, Dataset, Image, NerfTrainerConfig
    fields = {
        "Image": (Image, ["data", "width", "height", "channels"]),
        "Camera": (Camera, ["position", "rotation", "focal", "width", "height"]),
        "Dataset": (Dataset, ["images", "cameras", "num_images"]),
        "MLP": (MLP, [f[0] for f in MLP._fields_]),
        "NerfTrainerConfig": (NerfTrainerConfig, [f[0] for f in NerfTrainerConfig._fields_]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#define NERF_NO_HIP_RUNTIME 1']
    lines += [f'#include "{h}"' for h in ["nerf_data.h", "nerf_mlp.h", "nerf_amd.h"]]
    lines.append("int main(void) {")
    for name, (_, fl) in fields.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for f in fl:
            lines.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines.append("  return 0; }")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", INCLUDE, "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, (cls, fl) in fields.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for f in fl:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"


def test_missing_library_fails_loudly(tmp_path):
    """The Python front end has no fallback: without the shared object the import raises."""
    code = (
        "import importlib.util, sys, os\n"
        f"src = open(os.path.join({ROOT!r}, 'nerf_amd', '_lib.py')).read()\n"
        f"p = os.path.join({str(tmp_path)!r}, '_lib.py'); open(p, 'w').write(src)\n"
        "spec = importlib.util.spec_from_file_location('_lib_copy', p)\n"
        "m = importlib.util.module_from_spec(spec)\n"
        "try:\n    spec.loader.exec_module(m)\nexcept ImportError as e:\n    print('IMPORTERROR', e); sys.exit(0)\n"
        "sys.exit(1)\n"
    )
    out = subprocess.run(["python", "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "IMPORTERROR" in out.stdout, out.stdout + out.stderr
