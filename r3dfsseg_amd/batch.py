"""A batch of E episodes of one shape, resident on the device: the unit of work of the batched launch sequence
(head_train.explicit_train_batch, MPTI_SelfAtten.forward_episodes).

The reference runs one episode per step (DataLoader(batch_size=1), dataloaders/loader.py:1662-1684; `--batch_size` is
parsed and never used); episodes are independent units, so stacking them changes no per-episode result.
Tensors are the reference's per-episode tensors with a leading episode axis:
    support_x (E, n_way, k_shot, C, N) f32      support_y (E, n_way, k_shot, N) i32 {0,1}
    query_x   (E, n_q, C, N) f32                query_y   (E, n_q, N) i64
    gt_support_y, gt_query_y (train layout)     support_flag (E, n_way, k_shot) i32
x_all (E, S + Q, C, N) is what the encoder reads: per episode the S support clouds, then the Q query clouds.
Clouds that arrive as transposed views of point-major rows (episode_io's collate, as the reference's) stay that way: the
stacking copies rows, x_all is again such a view and the encoder reads it as it lies (ops.input_layouts)."""
import torch

from . import ops


class EpisodeBatch:
    def __init__(self, support_x, support_y, query_x, query_y, gt_support_y=None, gt_query_y=None, support_flag=None):
        self.E = support_x.shape[0]
        E = self.E
        pm = ops.is_point_major_view(support_x) and ops.is_point_major_view(query_x)
        self.support_x = support_x if pm else support_x.float().contiguous()
        self.support_y = support_y.to(torch.int32).contiguous()
        self.query_x = query_x if pm else query_x.float().contiguous()
        self.query_y = query_y.to(torch.int64).contiguous()
        self.gt_support_y = (gt_support_y if gt_support_y is not None else support_y).to(torch.int32).contiguous()
        self.gt_query_y = (gt_query_y if gt_query_y is not None else query_y).to(torch.int64).contiguous()
        self.support_flag = support_flag.to(torch.int32).contiguous() if support_flag is not None else None
        _, n_way, k_shot, C, N = self.support_x.shape
        if pm:
            self.x_all = ops.cat_clouds(self.support_x.reshape(E, n_way * k_shot, C, N), self.query_x, 1)
        else:
            self.x_all = torch.cat((self.support_x.view(E, n_way * k_shot, C, N), self.query_x), 1).contiguous()

    @staticmethod
    def from_episodes(episodes):
        """episodes: train-layout lists (loader.py:1666-1671: support_x, support_y, query_x, query_y, support_c, query_c,
        gt_support_y, gt_query_y, bg_pcd_x, bg_pcd_y, support_flag) or test-layout sequences starting with
        (support_x, support_y, query_x, query_y)."""
        def st(i):
            ts = [ep[i] for ep in episodes]
            if i in (0, 2) and all(ops.is_point_major_view(t) for t in ts):  # clouds as point-major views: stack the rows
                return torch.stack([t.transpose(-1, -2) for t in ts], 0).transpose(-1, -2)
            return torch.stack(ts, 0)
        full = len(episodes[0]) >= 11
        return EpisodeBatch(st(0), st(1), st(2), st(3), st(6) if full else None, st(7) if full else None,
                            st(10) if full else None)

    def __len__(self):
        return self.E

    def episode(self, e):
        """Episode e in the reference's train layout (views)."""
        return [self.support_x[e], self.support_y[e], self.query_x[e], self.query_y[e], None, None, self.gt_support_y[e],
                self.gt_query_y[e], None, None, self.support_flag[e] if self.support_flag is not None else None]
