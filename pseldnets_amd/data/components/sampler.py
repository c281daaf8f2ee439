"""Rank-strided global batch sampler — mirror of the reference's `data/components/sampler.py`
(UserDistributedBatchSampler :5-49): global batch = batch_size x world, rank r takes indices[p+r : p+G : world],
reshuffle with RandomState(seed) when an epoch is exhausted, last batch supplemented by wrap-around (a whole extra
batch when clip_num is already divisible — reproduced)."""
import numpy as np
import torch.distributed as dist


class UserDistributedBatchSampler:
    def __init__(self, clip_num, batch_size=1, seed=2023, data_indices=None, shuffle=True, last_batch_supplement=True):
        ready = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if ready else 0
        self.num_replicas = dist.get_world_size() if ready else 1
        self.batch_size = batch_size * self.num_replicas
        order = np.arange(clip_num) if data_indices is None else np.asarray(data_indices)
        self.clip_num = len(order)
        self.shuffle = shuffle
        self.pointer = 0
        if shuffle:
            self.random_state = np.random.RandomState(seed)
            self.random_state.shuffle(order)
        if last_batch_supplement:
            pad = self.batch_size - self.clip_num % self.batch_size
            order = np.append(order, order[:pad])
            self.clip_num += pad
        self.indices = order

    def __iter__(self):
        while True:
            if self.pointer >= self.clip_num:
                self.pointer = 0
                if self.shuffle:
                    self.random_state.shuffle(self.indices)
            lo = self.pointer
            self.pointer += self.batch_size
            yield self.indices[lo + self.rank: lo + self.batch_size: self.num_replicas]

    def __len__(self):
        return int(np.ceil(self.clip_num / self.batch_size))
