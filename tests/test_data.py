"""RASampler (cream_amd/autoformer/data.py) against index sequences drawn from the reference's own class
(AutoFormer/lib/samplers.py, tests/golden/make_golden.py `rasampler`), plus the properties the scheme exists for."""
from helpers import load_json


def test_rasampler_draws_the_reference_sequences():
    from cream_amd.autoformer.data import RASampler
    for rec in load_json("rasampler.json"):
        n, R, r, ep, sh = rec["case"]
        s = RASampler(range(n), num_replicas=R, rank=r, shuffle=sh)
        s.set_epoch(ep)
        assert len(s) == rec["len"]
        assert list(iter(s)) == rec["indices"], rec["case"]


def test_three_copies_of_a_sample_go_to_three_ranks():
    from cream_amd.autoformer.data import RASampler
    n, R = 1024, 4
    per_rank = []
    for r in range(R):
        s = RASampler(range(n), num_replicas=R, rank=r)
        s.set_epoch(3)
        per_rank.append(s.indices().tolist())
    assert all(len(p) == n // 256 * 256 // R for p in per_rank)
    first = {}                                                 # the first 3 consecutive positions of the global list
    for r, p in enumerate(per_rank):
        for k, idx in enumerate(p):
            first.setdefault(idx, set()).add((k * R + r) // 3 * 3)
    # every drawn sample comes from one triple of consecutive global positions, spread over ranks
    assert all(len(v) == 1 for v in first.values())
