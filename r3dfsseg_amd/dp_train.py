"""Data-parallel training step: episodes are sharded over ranks, every rank runs
forward + backward of its own episodes into ONE flat fp32 gradient bucket, a single RCCL all-reduce
(SUM) over xGMI follows and every rank applies the same Adam step.  The reference itself steps Adam
after every single episode (mpti_train_noise.py:57,98); with world_size 1 and one episode per step
this class does exactly that.  BatchNorm running statistics stay per rank (rank 0's are the ones
checkpointed), as documented in DESIGN.md."""
import torch

from . import dist as D


class DPTrainer:
    def __init__(self, learner, n_slots=0, example=None, lp_budget=None):
        """n_slots > 0: the local episodes of a step are replayed as captured hipGraphs, n_slots in flight on
        separate HIP streams (episode_graph.EpisodeGraphs; `example` = one episode fixing the shapes); every
        slot accumulates into its own gradient row and the rows are summed into the bucket before the
        all-reduce.  n_slots == 0: eager launches, one episode after the other."""
        self.learner = learner
        self.model = learner.model
        self.bucket = D.FlatGradBucket(self.model.parameters())
        self.graphs = None
        if n_slots:
            from .episode_graph import EpisodeGraphs
            self.rows = torch.zeros(n_slots, self.bucket.store.numel(), device=self.bucket.store.device)
            self.graphs = EpisodeGraphs(self.model, example, n_slots, train=True, lp_budget=lp_budget, grad_rows=self.rows)

    def step(self, episodes, logger=None):
        """episodes: list of train-layout data lists (loader.py:1666-1671) local to this rank.
        Returns the mean (lp_loss + 0.1 contrast) over the local episodes as a device tensor."""
        self.model.train()
        if self.graphs is not None:
            total = self.graphs.run(episodes)
            torch.sum(self.rows, 0, out=self.bucket.store)
            self.bucket.all_reduce_mean(len(episodes))
            self.learner.optimizer.step()
            self.learner.lr_scheduler.step()
            return total / max(len(episodes), 1)
        self.bucket.zero_()
        total = None
        for data in episodes:
            (support_x, support_y, query_x, query_y, support_c, query_c, gt_support_y, gt_query_y, bg_pcd_x, bg_pcd_y,
             support_flag) = data
            out = self.model(support_x, support_y, query_x, query_y, gt_support_y=gt_support_y, gt_query_y=gt_query_y,
                             train=True, logger=logger, support_flag=support_flag)
            loss = out[1] + 0.1 * out[2]  # mpti_learner.py:66
            loss.backward()               # accumulates into the bucket views
            total = loss.detach() if total is None else total + loss.detach()
        self.bucket.all_reduce_mean(len(episodes))
        self.learner.optimizer.step()
        self.learner.lr_scheduler.step()
        return total / max(len(episodes), 1)
