"""The C ABI from plain C99 (tests/c_abi/c_abi_smoke.c): compiled with gcc against include/blu_consensus.h, linked
against libblu_consensus.so, run as a child process.  CPU part: it builds, links, loads and a host-only handle refuses
to run.  GPU part: the records it gets for a C1-shaped table have the checksum of the oracle's records for the same
table (the table's LCG is restated below)."""
import os
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c")
LIBDIR = os.path.join(ROOT, "blutils_amd", "lib")


@pytest.fixture(scope="module")
def smoke_binary(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c_abi") / "c_abi_smoke")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), SRC,
                    "-L" + LIBDIR, "-lblu_consensus", "-Wl,-rpath," + LIBDIR, "-o", exe], check=True)
    return exe


def test_c_caller_builds_links_and_is_refused_without_a_device(smoke_binary):
    r = subprocess.run([smoke_binary, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "refused with BLU_ERR_NO_DEVICE" in r.stdout


def _table():
    """The LCG table of c_abi_smoke.c, restated."""
    from blutils_amd import synth
    M = (1 << 64) - 1
    state = [0xB10751]

    def rnd():
        state[0] = (state[0] * 6364136223846793005 + 1442695040888963407) & M
        return state[0] >> 33

    n_tax, depth, n_q, hpq = 2048, 8, 1000, 10
    div = [2048, 1024, 256, 64, 32, 8, 2, 1]
    t = np.arange(n_tax)
    node = np.stack([100000 * j + t // div[j] for j in range(depth)], axis=1).astype(np.uint32).reshape(-1)
    rank = np.tile(np.arange(depth, dtype=np.uint16), n_tax)
    tax = synth.SynthTaxonomy(["d", "k", "p", "c", "o", "f", "g", "s"], (np.arange(n_tax + 1) * depth).astype(np.uint64), node, rank,
                              t.astype(np.int64), np.zeros((9, n_tax), np.int32), np.zeros((9, n_tax), np.int32), n_tax, 0, False)
    masks = [0, 1, 7, 31, 63]
    H_ = n_q * hpq
    hits = {"seg_off": (np.arange(n_q + 1) * hpq).astype(np.int64), "bitscore": np.zeros(H_, np.int32), "tax_row": np.zeros(H_, np.int32),
            "pident": np.zeros(H_, np.float64), "align_len": np.zeros(H_, np.int32), "acc_rank": np.zeros(H_, np.uint32)}
    for q in range(n_q):
        anchor, mask, g = rnd() % n_tax, masks[rnd() % 5], 1 + rnd() % 4
        for j in range(hpq):
            r = q * hpq + j
            subject = (anchor & ~mask) + rnd() % (mask + 1)
            hits["bitscore"][r] = 500 if j < g else 500 - 1 - rnd() % 16
            hits["pident"][r] = (80000 + rnd() % 20001) / 1000.0
            hits["align_len"][r] = 380 + rnd() % 101
            hits["acc_rank"][r] = subject * 7 + 1
            hits["tax_row"][r] = subject
    return tax, hits


@pytest.mark.gpu
def test_c_caller_gets_the_oracles_records(smoke_binary):
    r = subprocess.run([smoke_binary], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    tax, hits = _table()
    exp = H.columnar(tax, hits, "custom", "relaxed", H.CUSTOM_16S, threads=2)
    fnv = 0xcbf29ce484222325
    for b in exp.tobytes():
        fnv = ((fnv ^ b) * 0x100000001b3) & ((1 << 64) - 1)
    assert f"checksum {fnv:016x}" in r.stdout, (r.stdout, f"{fnv:016x}")
    assert (exp["status"] == 0).sum() > 300 and (exp["status"] == 1).sum() > 100
