"""GPU parity of the neighbourhood-size kernel (acx_ball_sizes) with the reference's own program (golden sizes) and the
C oracle, both move sets; plus the edge cases of the reference's reader (zeros anywhere, unsorted input pair)."""
import pytest

from tests.conftest import ms_pool_generator_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ball():
    from ac_solver import _acx
    from ac_solver.barcode import neighbourhood_sizes

    _acx.require_device()
    return neighbourhood_sizes


def test_reference_golden_sizes(ball, golden_json):
    cases = golden_json("ball_sizes.json")["cases"]
    for classic in (False, True):
        for radius in (2, 4, 5):
            sel = [c for c in cases if c["classic"] == classic and c["radius"] == radius]
            by_width = {}
            for c in sel:
                by_width.setdefault(len(c["presentation"]), []).append(c)
            for group in by_width.values():
                got = ball([c["presentation"] for c in group], radius, classic)
                assert got == [c["size"] for c in group], (classic, radius, [c["tag"] for c in group])


def test_against_oracle_and_max_length(ball, golden_json):
    from oracle import ac_oracle as O

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    rows = [pool[k] for k in (1020, 1050, 1111, 1150)]  # n = 7: relators grow to 176 letters inside the radius-5 ball
    for classic in (False, True):
        sizes, longest = ball(rows, 5, classic, return_max_length=True)
        for r, s, m in zip(rows, sizes, longest):
            assert (s, m) == O.ball_size(r, 5, classic, return_max_length=True)
        assert max(longest) > 128


def test_reader_conventions_and_radius_zero(ball):
    from oracle import ac_oracle as O

    assert ball([1, 0, 0, 0, 2, 0, 0, 0], 0) == 1
    # zeros are dropped wherever they stand and the pair is sorted before the search (neibourhoods.cpp:76-88)
    a = ball([0, 1, 2, 0, 2, 0, 0, 1], 3)
    b = ball([2, 1, 0, 0, 1, 2, 0, 0], 3)
    assert a == b == O.ball_size([2, 1, 0, 0, 1, 2, 0, 0], 3)
    # inverse relators: the product is the empty word, which the reference keeps as a node
    assert ball([1, 2, 0, -2, -1, 0], 3) == O.ball_size([1, 2, 0, -2, -1, 0], 3)
    with pytest.raises(ValueError):
        ball([3, 0, 1, 0], 2)


def test_all_miller_schupp_presentations_radius_3(ball, golden_json):
    """every one of the 1190 presentations, both move sets, against the oracle at radius 3 (seconds on the CPU)"""
    from oracle import ac_oracle as O

    pool = ms_pool_generator_order(golden_json("ms_pool.json"))
    for n in range(7):
        rows = pool[n * 170:(n + 1) * 170]
        for classic in (False, True):
            got = ball(rows, 3, classic)
            want = [O.ball_size(r, 3, classic) for r in rows[::17]]
            assert got[::17] == want, (n, classic)
            assert min(got) > 100
