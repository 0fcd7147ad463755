"""every switch of the library's one table (csrc/knobs.h) is explained in DESIGN.md's switches table"""
import os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_switch_is_documented():
    names = re.findall(r"X\((ZV_[A-Z0-9_]+),", open(os.path.join(ROOT, "zerovox.cpp_amd", "csrc", "knobs.h")).read())
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(names) >= 30
    missing = [n for n in names if n not in design]
    assert not missing, missing


def test_no_launch_path_reads_the_environment():
    """the environment is read once, in knobs.cpp; no other source of the library calls getenv (ZEROVOX_DEVICE: the facade's device index)"""
    src = os.path.join(ROOT, "zerovox.cpp_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        if not f.endswith((".cpp", ".hip", ".h")) or f in ("knobs.cpp",):
            continue
        for i, ln in enumerate(open(os.path.join(src, f), errors="replace"), 1):
            if "getenv(" in ln and "ZEROVOX_DEVICE" not in ln and not ln.lstrip().startswith("//"):
                hits.append((f, i, ln.strip()))
    assert not hits, hits
