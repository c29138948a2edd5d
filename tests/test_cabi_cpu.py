"""The C-ABI library loads and exports every symbol include/vo_hip.h declares; the C++
drop-in layer builds and its GPU-free checks pass.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np

from visual_odometry_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared()
    assert len(names) >= 15
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vo_hip.h but not exported"
    assert sorted(capi.EXPORTS) == names


def test_struct_layouts_match_the_header(tmp_path):
    """Sizes and field offsets of the two records, as a C compiler lays out include/vo_hip.h."""
    assert capi.KP_DTYPE.itemsize == 28          # cv::KeyPoint layout
    fields = {"vo_keypoint": list(capi.KP_DTYPE.names), "vo_pair_result": list(capi.PAIR_DTYPE.names)}
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "vo_hip.h"', "int main(void) {"]
    for st, names in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for n in names:
            lines.append(f'printf("{st}.{n} %zu\\n", offsetof({st}, {n}));')
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, dt in (("vo_keypoint", capi.KP_DTYPE), ("vo_pair_result", capi.PAIR_DTYPE)):
        assert int(got[st]) == dt.itemsize
        for n in dt.names:
            assert int(got[f"{st}.{n}"]) == dt.fields[n][1], f"{st}.{n}"


def test_pattern_tables_agree():
    def load(p):
        s = re.sub(r"/\*.*?\*/", "", open(p).read(), flags=re.S)
        return [int(x) for x in s.replace("\n", " ").split(",") if x.strip()]
    a = load(os.path.join(ROOT, "oracle", "orb_pattern.inc"))
    b = load(os.path.join(ROOT, "visual_odometry_amd", "csrc", "orb_pattern.inc"))
    assert a == b and len(a) == 1024 and min(a) >= -13 and max(a) <= 13
    assert a[:16] == [8, -3, 9, 5, 4, 2, 7, -12, -11, 9, -8, 2, 7, -12, 12, -13] and a[-4:] == [-1, -6, 0, -11]


def test_cpp_dropin_builds_and_host_checks_pass():
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    assert os.path.exists(exe), "run `make` first"
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr


def test_product_never_touches_the_oracle():
    # the shipped package must not import, link or open anything under oracle/
    pkg = os.path.join(ROOT, "visual_odometry_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "synth.py", f"{f} mentions the oracle"
