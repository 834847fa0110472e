"""Evolution search: the host-side mirror of AutoFormer/evolution.py must visit exactly the
candidates the reference's own EvolutionSearcher visits (same CPython `random` draw order, same legality
rule, same selection) — fixture tests/golden/evolution_trace.json was produced by executing the reference's
class with a stubbed evaluator (tests/golden/make_golden.py evolution).  On the GPU the real evaluator runs."""
import json
import os
import random
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fixture_utils import SUPERNETS, model_kwargs  # noqa: E402
from make_golden import fake_accuracy  # noqa: E402  (pure function, no reference needed)


def test_search_visits_the_reference_sequence():
    from cream_amd.autoformer import evolution as EV
    from cream_amd.autoformer.supernet import Vision_TransformerSuper
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "evolution_trace.json")))
    model = Vision_TransformerSuper(**model_kwargs("S"))
    visited = []

    def evaluate(batches, config):
        if batches == "val":
            visited.append([config["layer_num"], config["mlp_ratio"], config["num_heads"], config["embed_dim"][0]])
        return {"acc1": fake_accuracy(config, batches)}

    a = fix["args"]
    es = EV.EvolutionSearcher(model, SUPERNETS["S"]["choices"], "val", "test", max_epochs=a["max_epochs"],
                              select_num=a["select_num"], population_num=a["population_num"], m_prob=a["m_prob"],
                              s_prob=a["s_prob"], crossover_num=a["crossover_num"], mutation_num=a["mutation_num"],
                              param_limits=a["param_limits"], min_param_limits=a["min_param_limits"], evaluate=evaluate)
    random.seed(0)
    top = es.search()
    as_lists = lambda x: json.loads(json.dumps(x))          # noqa: E731  (tuples -> lists, like the fixture)
    assert visited == fix["visited"]
    assert as_lists(es.memory) == fix["memory"] and as_lists(es.candidates) == fix["candidates"]
    assert as_lists(top) == fix["top50"] and as_lists(es.keep_top_k[a["select_num"]]) == fix["top_select"]
    assert es.top_accuracies == fix["top_accuracies"]
    for k, v in fix["params"].items():
        assert abs(es.vis_dict[eval(k)]["params"] - v) < 1e-9


def test_checkpoint_round_trip(tmp_path):
    from cream_amd.autoformer import evolution as EV
    from cream_amd.autoformer.supernet import Vision_TransformerSuper
    model = Vision_TransformerSuper(**model_kwargs("S"))
    ev = lambda b, c: {"acc1": fake_accuracy(c, b)}          # noqa: E731
    es = EV.EvolutionSearcher(model, SUPERNETS["S"]["choices"], "val", "test", output_dir=str(tmp_path), max_epochs=1,
                              select_num=2, population_num=4, crossover_num=1, mutation_num=1, evaluate=ev)
    random.seed(1)
    es.search()
    path = os.path.join(str(tmp_path), "checkpoint-1.pth.tar")      # evolution.py:58
    assert os.path.exists(path)
    es2 = EV.EvolutionSearcher(model, SUPERNETS["S"]["choices"], "val", "test", select_num=2, evaluate=ev)
    assert es2.load_checkpoint(path) and es2.epoch == 1 and es2.candidates == es.candidates
    assert set(torch.load(path, weights_only=False)) == {"top_accuracies", "memory", "candidates", "vis_dict", "keep_top_k", "epoch"}


@pytest.mark.gpu
def test_search_runs_on_the_native_evaluation_path():
    """Two small generations with the REAL evaluator (engine.evaluate: native block stack, bf16) on
    synthetic batches: every legal candidate is evaluated on the device, accuracies are finite, the best
    list is sorted."""
    from cream_amd.autoformer import engine, evolution as EV
    torch.manual_seed(0)
    model = engine.build_supernet("S", drop_path_rate=0.0).to("cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(3)
    batches = [(torch.randn(32, 3, 224, 224, device="cuda:0", generator=g),
                torch.randint(0, 1000, (32,), device="cuda:0", generator=g)) for _ in range(2)]
    es = EV.EvolutionSearcher(model, engine.SEARCH_SPACES["S"]["choices"], batches, batches[:1], max_epochs=2, select_num=3,
                              population_num=6, crossover_num=2, mutation_num=2, param_limits=40, min_param_limits=5)
    random.seed(0)
    top = es.search()
    accs = [es.vis_dict[c]["acc"] for c in top]
    assert es.evaluated >= 6 and all(a == a and 0.0 <= a <= 100.0 for a in accs) and accs == sorted(accs, reverse=True)
