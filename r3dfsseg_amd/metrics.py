"""Device-side mIoU accumulator replacing the per-point python triple loop of
eval_noise.py:23-72 (evaluate_metric): TP / GT / P histograms per test class, mean IoU over the
foreground classes only (eval_noise.py:70)."""
import torch

from . import ops


class MIoUAccumulator:
    def __init__(self, test_classes, device="cuda"):
        self.test_classes = [int(c) for c in test_classes]
        self.n_classes = len(self.test_classes) + 1
        self.hist = torch.zeros(3, self.n_classes, dtype=torch.int64, device=device)

    def update(self, pred, gt, label2class):
        """pred / gt (n_q, N) episode labels in 0..n_way; label2class = sampled_classes of the episode."""
        lut = [0] + [self.test_classes.index(int(c)) + 1 for c in label2class]
        lut = torch.tensor(lut, dtype=torch.int32, device=self.hist.device)
        ops.miou_accumulate(pred, gt, lut, self.hist)

    def reduce(self):
        from . import dist as D
        D.all_reduce_histogram(self.hist)

    def compute(self):
        gt, pos, tp = [h.double() for h in self.hist.cpu()]
        iou = tp / (gt + pos - tp)
        return float(iou[1:].mean()), iou.numpy()
