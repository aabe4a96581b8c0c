"""CPU: the build-time tunables of sdfest_amd/csrc/tuning.hpp are numbers a variant build may override
(tools/microbench/build_variant.sh).  hipcc cross-compiles gfx950 without a GPU: a build with every one of them
changed must still compile (device code, syntax + semantics), so the harness's knobs cannot rot."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdfest_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

OVERRIDES = {"SDFR_MACRO_SX": "2", "SDFR_MACRO_SY": "1", "SDFR_FWD_SX": "4", "SDFR_FWD_SY": "1",
             "SDFR_BWD_MACRO_MIN": "8192", "SDFR_FWD_MACRO_MIN": "8192", "SDFR_FWD_WIDE": "0", "SDFR_FWD_WAVES": "4",
             "SDFR_BWD_BIG_MIN_RATIO": "2.4f", "SDFR_PACKED_MIN_VIEWS": "2", "SDFR_BWD_SLOTS": "256",
             "SDFR_DENSE_CAP": "3072", "SDFR_BWD_WAVES_PER_EU": "0", "SDFR_BT_LOADS": "6", "SDFR_BT_OUT": "4",
             "SDFR_BT_TILE": "4", "SDFR_BT_THREADS": "512", "SDFR_PC_GRID_TARGET": "2048", "SDFR_PC_MIN_GROUPS": "16",
             "SDFR_DIRECT_MIN_TILES": "128", "SDFR_DIRECT_MIN_TILES_FEW": "64", "SDFR_DIRECT_MIN_LATENTS_FEW": "4", "SDFR_FEW_MIX_LOG2": "20", "SDFR_RESIZE_TILED_MIN_ITEMS": "1024", "SDFR_INLINE_MAX_VIEWS": "4", "SDFR_SPLITK_MAX_LATENTS": "8",
             "SDFR_SMALL_DIRECT": "1", "SDFR_FUSED_MAX_VIEWS": "4", "SDFR_FUSED_DIRECT_MAX_POINTS": "1000"}


def test_every_tunable_is_listed_and_overridable():
    text = open(os.path.join(CSRC, "tuning.hpp")).read()
    names = set(re.findall(r"#ifndef (SDFR_\w+)", text)) | {"SDFR_MACRO_SY", "SDFR_FWD_SY"}
    assert names == set(OVERRIDES), names ^ set(OVERRIDES)
    # no other override points hide in the sources
    for f in os.listdir(CSRC):
        if f == "tuning.hpp" or not f.endswith((".hip", ".hpp", ".cpp")):
            continue
        hidden = re.findall(r"#ifndef (SDFR_\w+)", open(os.path.join(CSRC, f)).read())
        assert hidden in ([], ["SDFR_SPLITK_MAX_TILES"]), (f, hidden)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_a_build_with_every_tunable_changed_compiles():
    flags = [f"-D{k}={v}" for k, v in OVERRIDES.items()]
    for src in ("render.hip", "sampler.hip", "decoder.hip"):
        cmd = [HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only",
               f"-I{os.path.join(ROOT, 'include')}", "-x", "hip", os.path.join(CSRC, src)] + flags
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
