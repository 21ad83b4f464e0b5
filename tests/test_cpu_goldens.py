"""The committed oracle outputs (tests/golden/llm_goldens.npz, written by tests/golden/make_llm_goldens.py) must be reproduced bit for bit by the oracle as
built now, from model files regenerated from their seeds: pins the synthetic-model generator, the quantisers used to write the files and the CPU oracle
against silent drift.  (They are goldens of the oracle, not of ggml -- DESIGN.md section 2.)"""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_llm_goldens as M  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_goldens.npz"))


@pytest.mark.parametrize("wtype,mix", M.CASES)
def test_oracle_llm_reproduces_goldens(tmp_path, wtype, mix):
    sha, logits, ids, margins, last = M.llm_case(wtype, mix, str(tmp_path))
    assert sha == str(GOLD[f"{wtype}/file_sha256"]), "the synthetic model file changed: generator or quantiser drift"
    assert np.array_equal(logits, GOLD[f"{wtype}/prompt_logits"])
    assert np.array_equal(ids, GOLD[f"{wtype}/greedy_ids"])
    assert np.array_equal(last, GOLD[f"{wtype}/final_logits"])


def test_oracle_vision_reproduces_goldens(tmp_path):
    sha, emb = M.vision_case(str(tmp_path))
    assert sha == str(GOLD["vision/file_sha256"])
    assert np.array_equal(emb[:4], GOLD["vision/embedding_rows_0_3"])
    assert hashlib.sha256(np.ascontiguousarray(emb).tobytes()).hexdigest() == str(GOLD["vision/embedding_sha256"])
