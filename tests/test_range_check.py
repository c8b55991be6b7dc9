"""The lazily reduced sums of fr.h / poseidon.h / the signature ladders stay inside the ranges their routines state: proved by
interval arithmetic with the real constants (tools/range_check.py; VERDICT r3 "parity gaps" 1c), and the proof is tied to the
source text it transcribes."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import range_check as RC  # noqa: E402


def test_every_bound_holds_and_the_transcription_is_current():
    rep = []
    missing = RC.run(rep)
    assert missing == [], "tools/range_check.py quotes source that changed: %r" % missing[:3]
    assert len(rep) == 12 + 4 + 2 + 1
    # the partial-round lanes really come close to the limit the trim interval was chosen for: the check is not vacuous
    peaks = [float(l.split("peak at ")[1].split(" p")[0]) for l in rep if "peak at" in l]
    assert len(peaks) == 12 and all(7.0 < x < 8.0 for x in peaks)


@pytest.mark.parametrize("T", [3, 5])
def test_a_longer_trim_interval_is_rejected(T):
    """trimming the lanes after every FOURTH pair would hand fr_muladd2 an addend above 8p: the checker must say so"""
    with pytest.raises(RC.RangeError) as e:
        RC.poseidon_hash(T, True, trim_every=4)
    assert "8p" in str(e.value)


def test_an_unreduced_operand_is_rejected():
    big = RC.V(11 * RC.P)   # 2^257 is 10.6 p
    with pytest.raises(RC.RangeError):
        RC.fr_mul(big, RC.V(RC.P))
    with pytest.raises(RC.RangeError):
        RC.fr_is_zero(RC.V(3 * RC.P))
    with pytest.raises(RC.RangeError):
        RC.fr_dot([RC.V(RC.P)] * 7, [RC.V(RC.P)] * 7)
