"""CPU: the headline numbers the documents quote are the committed bench line's (round-5 verdict: README said
"~700x" and "0.457" where the driver measured 442.5x and 0.452).  README.md and DESIGN.md section 5 each carry ONE
sentence of the form

    Headline (`profiles/bench_r06.json`): **977 k renders/s** forward+backward ... **0.452 of the HBM roofline** ... **442x the CPU port** ...

and this test fails when any of the three differs from profiles/bench_r06.json by more than 3 %."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "profiles", "bench_r06.json")
PATTERN = re.compile(r"Headline \(`profiles/bench_r06\.json`\): \*\*([\d.]+) k renders/s\*\*.*?\*\*([\d.]+) of the HBM "
                     r"roofline\*\*.*?\*\*([\d.]+)x the CPU port\*\*", re.S)


def bench_line():
    with open(BENCH) as f:
        text = f.read().strip()
    return json.loads(text.splitlines()[-1])


@pytest.mark.parametrize("doc", ["README.md", "DESIGN.md"])
def test_headline_sentence_matches_the_committed_bench_line(doc):
    line = bench_line()
    with open(os.path.join(ROOT, doc)) as f:
        m = PATTERN.search(f.read())
    assert m, f"{doc}: no headline sentence of the agreed form"
    renders_k, frac, speedup = (float(x) for x in m.groups())
    assert abs(renders_k * 1e3 / line["value"] - 1) <= 0.03, (renders_k, line["value"])
    assert abs(frac / line["roofline_step"]["frac"] - 1) <= 0.03, (frac, line["roofline_step"]["frac"])
    assert abs(speedup / line["speedup_vs_cpu"] - 1) <= 0.03, (speedup, line["speedup_vs_cpu"])


def test_the_committed_line_is_a_default_run():
    line = bench_line()
    assert line["n_gpus"] == 1 and line["unit"] == "renders/s" and line["dtype"] == "f32"
    assert {"roofline", "cpu_baseline", "parity", "configs"} <= set(line)
