"""Evolutionary sub-network search over a trained supernet — host-side mirror of
AutoFormer/evolution.py:18-290 on the native evaluation path.

A candidate is the reference's flat tuple  (depth, mlp_ratio x depth, num_heads x depth, embed_dim)
(`decode_cand_tuple`, evolution.py:18-20).  One generation (evolution.py:238-276):
  * fill the population with random legal candidates (`get_random`, :143-152),
  * keep the `select_num` best by validation top-1 (and a top-50 list) (`update_top_k`, :112-118),
  * produce `mutation_num` mutants (:154-207: depth re-draw with probability s_prob, per-layer re-draws
    with probability m_prob, embed-dim re-draw with probability s_prob) and `crossover_num` children
    (:209-236: per-position choice between two parents of equal length),
  * the next population = mutants + children, topped up with fresh random candidates.
A candidate is legal once (`is_legal`, :77-110): not visited before, parameter count of the sampled
sub-network (get_sampled_params_numel) inside [min_param_limits, param_limits] million, and it is then
EVALUATED — that is the hot part: every legal candidate costs a pass over the validation and test
batches through `engine.evaluate` (the native block stack: one C call per block, no host sync inside).

The order of CPython `random` draws is the reference's (candidates are generated in batches of 10 by
`stack_random_cand`, :120-128, whether or not all ten are consumed), so with the same seed and the same
accuracies the search visits the same candidates: tests/test_evolution.py pins the visited sequence
against a fixture produced by executing the reference's own class with a stubbed evaluator.
"""
import os
import random

import torch

from . import engine


def decode_cand_tuple(cand):
    depth = cand[0]
    return depth, list(cand[1:depth + 1]), list(cand[depth + 1:2 * depth + 1]), cand[-1]


def config_of(cand):
    depth, mlp_ratio, num_heads, embed_dim = decode_cand_tuple(cand)
    return dict(layer_num=depth, mlp_ratio=mlp_ratio, num_heads=num_heads, embed_dim=[embed_dim] * depth)


class EvolutionSearcher:
    def __init__(self, model, choices, val_batches, test_batches, output_dir=None, max_epochs=20, select_num=10,
                 population_num=50, m_prob=0.2, s_prob=0.4, crossover_num=25, mutation_num=25, param_limits=23.0,
                 min_param_limits=18.0, amp_dtype=torch.bfloat16, evaluate=None, log=None):
        self.model, self.choices = model, choices
        self.val_batches, self.test_batches = val_batches, test_batches
        self.output_dir = output_dir
        self.max_epochs, self.select_num, self.population_num = max_epochs, select_num, population_num
        self.m_prob, self.s_prob = m_prob, s_prob
        self.crossover_num, self.mutation_num = crossover_num, mutation_num
        self.parameters_limits, self.min_parameters_limits = param_limits, min_param_limits
        self.amp_dtype = amp_dtype
        self._evaluate = evaluate or self._evaluate_native
        self.log = log or (lambda *a: None)
        self.memory, self.vis_dict = [], {}
        self.keep_top_k = {self.select_num: [], 50: []}
        self.epoch = 0
        self.candidates, self.top_accuracies = [], []
        self.evaluated = 0
        self._pending = []                     # (info, key, PendingEval): accuracies still on the device

    # ---- the hot part ------------------------------------------------------------------------
    def _evaluate_native(self, batches, config):
        # deferred: the accuracy of a candidate is first READ when its population is ranked (update_top_k), so the candidates of a
        # phase are enqueued back to back and resolved together — the device never waits for the host between candidates
        return engine.evaluate(batches, self.model, amp_dtype=self.amp_dtype, mode='retrain', retrain_config=config, defer=True)

    def _record(self, info, key, res):
        if not isinstance(res, dict) and hasattr(res, 'result'):             # engine.PendingEval (a custom `evaluate` may return the dict)
            info[key] = None
            self._pending.append((info, key, res))
            if len(self._pending) >= 256:                                     # (a bound on what is outstanding, far above a population)
                self.resolve()
        else:
            info[key] = res['acc1']

    def resolve(self):
        """Bring every outstanding accuracy to the host (one wait for the device, then plain reads)."""
        for info, key, res in self._pending:
            info[key] = res.result()['acc1']
        self._pending = []

    def is_legal(self, cand):
        assert isinstance(cand, tuple)
        info = self.vis_dict.setdefault(cand, {})
        if 'visited' in info:
            return False
        config = config_of(cand)
        info['params'] = self.model.get_sampled_params_numel(config) / 10. ** 6
        if info['params'] > self.parameters_limits or info['params'] < self.min_parameters_limits:
            return False
        self._record(info, 'acc', self._evaluate(self.val_batches, config))
        self._record(info, 'test_acc', self._evaluate(self.test_batches, config))
        info['visited'] = True
        self.evaluated += 1
        return True

    # ---- candidate generation (draw order of the reference) --------------------------------------
    def stack_random_cand(self, random_func, batchsize=10):
        while True:
            cands = [random_func() for _ in range(batchsize)]
            for cand in cands:
                self.vis_dict.setdefault(cand, {})
            for cand in cands:
                yield cand

    def get_random_cand(self):
        depth = random.choice(self.choices['depth'])
        cand = [depth]
        for dimension in ('mlp_ratio', 'num_heads'):
            cand += [random.choice(self.choices[dimension]) for _ in range(depth)]
        cand.append(random.choice(self.choices['embed_dim']))
        return tuple(cand)

    def get_random(self, num):
        it = self.stack_random_cand(self.get_random_cand)
        while len(self.candidates) < num:
            cand = next(it)
            if self.is_legal(cand):
                self.candidates.append(cand)

    def _mutant(self, k):
        depth, mlp_ratio, num_heads, embed_dim = decode_cand_tuple(list(random.choice(self.keep_top_k[k])))
        if random.random() < self.s_prob:                                    # depth
            new_depth = random.choice(self.choices['depth'])
            if new_depth > depth:
                mlp_ratio = mlp_ratio + [random.choice(self.choices['mlp_ratio']) for _ in range(new_depth - depth)]
                num_heads = num_heads + [random.choice(self.choices['num_heads']) for _ in range(new_depth - depth)]
            else:
                mlp_ratio, num_heads = mlp_ratio[:new_depth], num_heads[:new_depth]
            depth = new_depth
        for values, name in ((mlp_ratio, 'mlp_ratio'), (num_heads, 'num_heads')):
            for i in range(depth):
                if random.random() < self.m_prob:
                    values[i] = random.choice(self.choices[name])
        if random.random() < self.s_prob:                                    # embed_dim
            embed_dim = random.choice(self.choices['embed_dim'])
        return tuple([depth] + mlp_ratio + num_heads + [embed_dim])

    def _child(self, k):
        p1, p2 = random.choice(self.keep_top_k[k]), random.choice(self.keep_top_k[k])
        tries = 50
        while len(p1) != len(p2) and tries > 0:
            tries -= 1
            p1, p2 = random.choice(self.keep_top_k[k]), random.choice(self.keep_top_k[k])
        return tuple(random.choice([i, j]) for i, j in zip(p1, p2))

    def _collect(self, make, want):
        res, budget = [], want * 10
        it = self.stack_random_cand(make)
        while len(res) < want and budget > 0:
            budget -= 1
            cand = next(it)
            if self.is_legal(cand):
                res.append(cand)
        return res

    def get_mutation(self, k, mutation_num):
        return self._collect(lambda: self._mutant(k), mutation_num)

    def get_crossover(self, k, crossover_num):
        return self._collect(lambda: self._child(k), crossover_num)

    def update_top_k(self, candidates, k, key, reverse=True):
        self.resolve()
        t = self.keep_top_k[k]
        t += candidates
        t.sort(key=key, reverse=reverse)
        self.keep_top_k[k] = t[:k]

    # ---- checkpoints: evolution.py:51-75 ('checkpoint-{epoch}.pth.tar') ----------------------------
    def state(self):
        self.resolve()
        return dict(top_accuracies=self.top_accuracies, memory=self.memory, candidates=self.candidates,
                    vis_dict=self.vis_dict, keep_top_k=self.keep_top_k, epoch=self.epoch)

    def save_checkpoint(self):
        if self.output_dir is None:
            return None
        path = os.path.join(self.output_dir, f"checkpoint-{self.epoch}.pth.tar")
        torch.save(self.state(), path)
        return path

    def load_checkpoint(self, path):
        if not os.path.exists(path):
            return False
        info = torch.load(path, weights_only=True)         # tuples / dicts / lists of python scalars: the safe unpickler reads them
        self.memory, self.candidates, self.vis_dict = info['memory'], info['candidates'], info['vis_dict']
        self.keep_top_k, self.epoch = info['keep_top_k'], info['epoch']
        self.top_accuracies = info.get('top_accuracies', [])
        return True

    # ---- the search loop: evolution.py:238-276 ---------------------------------------------------
    def search(self):
        self.get_random(self.population_num)
        while self.epoch < self.max_epochs:
            self.memory.append(list(self.candidates))
            acc = lambda x: self.vis_dict[x]['acc']            # noqa: E731
            self.update_top_k(self.candidates, k=self.select_num, key=acc)
            self.update_top_k(self.candidates, k=50, key=acc)
            self.top_accuracies.append([self.vis_dict[c]['acc'] for c in self.keep_top_k[50]])
            best = self.keep_top_k[50][0]
            self.log(f"epoch {self.epoch}: best {best} val {self.vis_dict[best]['acc']:.3f} "
                     f"params {self.vis_dict[best]['params']:.2f}M, {self.evaluated} sub-networks evaluated")
            mutation = self.get_mutation(self.select_num, self.mutation_num)
            crossover = self.get_crossover(self.select_num, self.crossover_num)
            self.candidates = mutation + crossover
            self.get_random(self.population_num)
            self.epoch += 1
            self.save_checkpoint()
        self.resolve()
        return self.keep_top_k[50]
