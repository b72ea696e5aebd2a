"""Register / LDS / scratch budgets of the kernels whose residency the launchers count on (CPU: hipcc cross-compiles gfx950 and
reports per kernel with -Rpass-analysis=kernel-resource-usage).  A couple of registers more can silently halve the number of
resident waves -- k_embed_fwd<8> went from four waves per SIMD to three that way in round 3 -- so the intended figures are pinned:
  * no kernel spills to scratch;
  * k_embed_fwd<8>: four waves per SIMD (two 8-wave blocks per CU on large row sets, launch_embed_fwd);
  * k_wgrad: two waves per SIMD (ONE resident round of two blocks per CU, place_wg);
  * k_reduce: at least six waves per SIMD and at most 26 KB of LDS (six blocks per CU: one round for ~1,200 blocks);
  * the edge passes without the long-segment body: at least six waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("res") / "k.s"
    src = os.path.join(ROOT, "gcnn-cut-selector_amd", "csrc", "gcnn_capi.hip")
    p = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", str(out), src,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rows, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    filt = shutil.which("c++filt")
    names = list(rows)
    if filt:
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        rows = {d: rows[n] for n, d in zip(names, dem)}
    return {k: v for k, v in rows.items() if re.search(r"\bk_[a-z_0-9]+", k)}


def _one(usage, prefix):
    hits = [v for k, v in usage.items() if k.replace("void ", "").startswith(prefix)]
    assert len(hits) == 1, (prefix, [k for k in usage if prefix.split("<")[0] in k])
    return hits[0]


def test_no_kernel_spills(usage):
    assert len(usage) > 60
    spilled = {k: v["scratch"] for k, v in usage.items() if v.get("scratch", 0)}
    assert not spilled, spilled


def test_pinned_residency(usage):
    assert _one(usage, "k_embed_fwd<8>")["occ"] >= 4
    wg = _one(usage, "k_wgrad(")
    assert wg["occ"] == 2 and wg["vgpr"] <= 256
    rd = _one(usage, "k_reduce(")
    assert rd["occ"] >= 6 and rd["lds"] <= 26 * 1024
    for k, v in usage.items():
        name = k.replace("void ", "")
        if name.startswith("k_edge_fwd<") and name.split("(")[0].endswith("false>"):
            assert v["occ"] >= 6, (k, v)
        if name.startswith("k_edge_bwd_send<") and name.split("(")[0].endswith("false>"):
            assert v["occ"] >= 6, (k, v)
