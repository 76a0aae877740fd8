"""Host-side mirror of the reference's models/protonet.py::ProtoNet (lines 39-58, 245-354):
same constructor and forward() signature; encoder / attention / base learner run on the HIP
kernels, the head is r3d_protonet_head."""
import torch
import torch.nn as nn

from . import ops
from .dgcnn import DGCNN, BaseLearner, SelfAttention


class ProtoNet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.n_way = args.n_way
        self.k_shot = args.k_shot
        self.dist_method = args.dist_method
        self.in_channels = args.pc_in_dim
        self.n_points = args.pc_npts
        self.use_attention = args.use_attention
        if self.n_way > 7:
            raise NotImplementedError("n_way <= 7 (the head kernels carry at most 8 classes)")
        self.encoder = DGCNN(args.edgeconv_widths, args.dgcnn_mlp_widths, args.pc_in_dim, k=args.dgcnn_k)
        self.base_learner = BaseLearner(args.dgcnn_mlp_widths[-1], args.base_widths)
        if self.use_attention:
            self.att_learner = SelfAttention(args.dgcnn_mlp_widths[-1], args.output_dim)
        else:
            self.linear_mapper = nn.Conv1d(args.dgcnn_mlp_widths[-1], args.output_dim, 1, bias=False)
        self.feat_dim = args.edgeconv_widths[0][-1] + args.output_dim + args.base_widths[-1]

    def getFeatures_pm(self, x):
        B, _, N = x.shape
        x_pm, x_cm = ops.input_layouts(x)
        cat, level2 = self.encoder.forward_pm(x_pm, B, N, x_cm=x_cm)
        feat = torch.empty(B * N, self.feat_dim, device=x.device, dtype=torch.float32)
        ops.copy_cols(cat[:, :64], feat[:, :64])
        if self.use_attention:
            self.att_learner.forward_pm(level2, B, N, feat[:, 64:128])
        else:
            W = self.linear_mapper.weight.reshape(64, -1).contiguous()
            ops.pointwise_conv(level2, W, None, None, ops.ACT_NONE, out=feat[:, 64:128])
        self.base_learner.forward_pm(level2, feat[:, 128:])
        return feat

    def getFeatures(self, x):
        B, _, N = x.shape
        return ops.pm_to_cm(self.getFeatures_pm(x), B, N)

    def forward(self, support_x, support_y, query_x, query_y, support_c=None, query_c=None, train=False,
                gt_support_y=None, gt_query_y=None, logger=None):
        if train or self.training:
            raise NotImplementedError("ProtoLearner.train is broken in the reference itself "
                                      "(proto_learner.py:57 unpacks 6 values from a forward that returns 2)")
        S, N = self.n_way * self.k_shot, self.n_points
        n_q = query_x.shape[0]
        sx = support_x.reshape(S, self.in_channels, N)
        feat = self.getFeatures_pm(torch.cat((sx, query_x), 0))
        Z = ops.protonet_head(feat[:S * N], feat[S * N:], support_y, self.n_way, self.k_shot, N, self.dist_method)
        labels = query_y.to(torch.int64).contiguous() if query_y is not None else None
        logits, loss, _ = ops.logits_ce_from_rows(Z, n_q, N, self.n_way + 1, labels)
        return logits, loss
