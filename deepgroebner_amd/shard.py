"""Placement of a global batch of independent environments on ranks (one process per GPU).

Environments never exchange data, so the only multi-GPU logic is this id/seed plan: global
environment g gets ideal seed 1000+g and agent seed g wherever it runs, rank r owns the contiguous
block [r*B, (r+1)*B).  No collective touches the data path."""
import numpy as np

IDEAL_SEED0 = 1000


def plan(rank, world, per_rank_batch):
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    ids = np.arange(per_rank_batch, dtype=np.int64) + rank * per_rank_batch
    return {"ids": ids, "ideal_seeds": ids + IDEAL_SEED0, "agent_seeds": ids.astype(np.uint32)}
