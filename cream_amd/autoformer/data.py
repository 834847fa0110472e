"""Host side of the input path next to the train step — mirror of AutoFormer/lib/samplers.py:6-57 (`RASampler`, the
repeated-augmentation sampler the reference trains with: `--repeated-aug` is its default, supernet_train.py:118-122,
209-213).

Same constructor, `set_epoch`, `__len__` and, for every (dataset length, replicas, rank, epoch, shuffle), the SAME index
sequence: each sample appears three times in a row in the epoch's permutation, the list is padded with its own head to a
multiple of the replica count, rank r takes every `num_replicas`-th entry from r (so the three copies of a sample go to
three different ranks, where they get three different augmentations), and an epoch is cut to
floor(len // 256 * 256 / replicas) draws.  The reference materialises the 3 x len Python list on every rank and epoch
(3.8 M boxed ints for ImageNet); here the rank's entries are computed directly as a tensor expression of the permutation.
"""
import math

import torch
import torch.distributed as dist


class RASampler(torch.utils.data.Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size()
        if rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            rank = dist.get_rank()
        self.dataset, self.num_replicas, self.rank, self.epoch, self.shuffle = dataset, num_replicas, rank, 0, shuffle
        n = len(dataset)
        self.num_samples = int(math.ceil(n * 3.0 / num_replicas))
        self.total_size = self.num_samples * num_replicas
        self.num_selected_samples = int(math.floor(n // 256 * 256 / num_replicas))

    def indices(self):
        """The rank's draws of this epoch as an int64 tensor."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        # position t of the repeated, padded list holds perm[(t mod 3n) // 3]; this rank reads t = rank, rank + R, ...
        t = torch.arange(self.rank, self.total_size, self.num_replicas)[:self.num_selected_samples]
        return perm[(t % (3 * n)) // 3] if n > 0 else perm

    def __iter__(self):
        return iter(self.indices().tolist())

    def __len__(self):
        return self.num_selected_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
