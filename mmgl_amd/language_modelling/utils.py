"""Trainer-side helpers with the reference's names (language_modelling/utils.py): parameter counting, meters whose
cross-rank mean is one all-reduce of [sum, count] (:113-118), HF feature-extractor helpers (:15-23)."""
from enum import Enum

import torch
import torch.distributed as dist


def get_feature_extractor_for_model(model_name: str):
    from transformers import AutoFeatureExtractor
    print(f"Using HuggingFace AutoFeatureExtractor for {model_name}.")
    return AutoFeatureExtractor.from_pretrained(model_name)


def get_pixel_values_for_model(feature_extractor, img):
    return feature_extractor(img.convert("RGB"), return_tensors="pt").pixel_values[0, ...]


def get_params_count(model, max_name_len: int = 60):
    """Returns (rows, trainable, non_trainable) like the reference (:26-30)."""
    rows = [(n[:max_name_len], p.numel(), "{}".format(list(p.size())), p.requires_grad) for n, p in model.named_parameters()]
    train = sum(r[1] for r in rows if r[3])
    frozen = sum(r[1] for r in rows if not r[3])
    return rows, train, frozen


class Summary(Enum):
    NONE = 0
    AVERAGE = 1
    SUM = 2
    COUNT = 3


class AverageMeter:
    """Running value / average; all_reduce() sums [sum, count] across ranks (reference :93-137)."""

    def __init__(self, name, fmt=":f", summary_type=Summary.AVERAGE):
        self.name, self.fmt, self.summary_type = name, fmt, summary_type
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        val = float(val)
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def all_reduce(self):
        if not (dist.is_available() and dist.is_initialized()):
            return
        device = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
        total = torch.tensor([self.sum, self.count], dtype=torch.float32, device=device)
        dist.all_reduce(total, dist.ReduceOp.SUM, async_op=False)
        self.sum, self.count = total.tolist()
        self.avg = self.sum / max(self.count, 1e-12)

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)

    def summary(self):
        key = {Summary.NONE: None, Summary.AVERAGE: "avg", Summary.SUM: "sum", Summary.COUNT: "count"}.get(self.summary_type, "bad")
        if key == "bad":
            raise ValueError("invalid summary type %r" % self.summary_type)
        return "" if key is None else "{} {:.3f}".format(self.name, getattr(self, key))


class ProgressMeter:
    def __init__(self, num_batches, meters, prefix=""):
        digits = len(str(num_batches // 1))
        self.batch_fmtstr = "[{:" + str(digits) + "d}/" + ("{:" + str(digits) + "d}").format(num_batches) + "]"
        self.meters, self.prefix = meters, prefix

    def display(self, batch):
        print("\t".join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters]))

    def display_summary(self):
        print(" ".join([" *"] + [m.summary() for m in self.meters]))
