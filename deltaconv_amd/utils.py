"""Training-step glue of the reference's experiment scripts (experiments/utils.py:7-51): the
label-smoothed cross entropy and the ShapeNet part-IoU metric.  Product code (runs on whatever device
the logits live on); the oracle keeps its own restatement for the CPU side of the parity tests."""
import numpy as np
import torch
import torch.nn.functional as F


class _CELoss(torch.autograd.Function):
    """loss and d loss / d logits from one pass (csrc/loss.hip): two launches instead of ~25."""

    @staticmethod
    def forward(ctx, pred, true, eps):
        from ._lib import lib
        pred = pred if (pred.dtype == torch.float32 and pred.stride(-1) == 1) else pred.contiguous().float()
        r, c = pred.shape
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dlogits = torch.empty(r, c, dtype=torch.float32, device=pred.device)
        nbytes = lib.raw("dc_ce_loss_workspace_bytes")(r)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=pred.device)
        lib.call("dc_ce_loss", pred, pred.stride(0), true, r, c, float(eps), loss, dlogits, c, ws, nbytes)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        if getattr(g, "_dc_unit_seed", False):      # graph_step.py seeds the backward pass with a tensor it KNOWS to be 1.0:
            return dlogits, None, None              # no scaling launch (a marked tensor is never written after creation)
        return dlogits * g, None, None


def calc_loss(pred, true, smoothing=True):
    """experiments/utils.py:7-24: cross entropy with label smoothing eps = 0.2 (classification) or
    plain mean cross entropy (segmentation, smoothing=False).  Logits on the GPU go through the fused
    HIP kernel; the torch formula below serves host-side tensors (metrics on CPU copies, CPU tests)."""
    true = true.contiguous().view(-1)
    if pred.is_cuda:
        assert pred.dim() == 2 and true.numel() == pred.shape[0], "calc_loss: pred [R,C], true [R]"
        return _CELoss.apply(pred, true.long(), 0.2 if smoothing else 0.0)
    if not smoothing:
        return F.cross_entropy(pred, true, reduction='mean')
    eps, n_class = 0.2, pred.size(1)
    logp = F.log_softmax(pred, dim=1)
    # sum_c q_c * logp_c with q = (1-eps) on the label and eps/(n_class-1) elsewhere
    picked = logp.gather(1, true.view(-1, 1)).squeeze(1)
    rest = logp.sum(dim=1) - picked
    return -((1 - eps) * picked + eps / (n_class - 1) * rest).mean()


def calc_shape_IoU(pred_np, seg_np, label, class_choice):
    """experiments/utils.py:27-51: mean part IoU per ShapeNet shape (an empty union counts as 1)."""
    seg_num = [4, 2, 2, 4, 4, 3, 3, 2, 4, 2, 6, 2, 3, 3, 3, 3]
    index_start = [0, 4, 6, 8, 12, 16, 19, 22, 24, 28, 30, 36, 38, 41, 44, 47]
    label = np.asarray(label).squeeze()
    ious = []
    for s in range(seg_np.shape[0]):
        if not class_choice:
            cat = int(label[s]) if label.ndim else int(label)
            parts = range(index_start[cat], index_start[cat] + seg_num[cat])
        else:
            parts = range(seg_num[int(label[0]) if label.ndim else int(label)])
        per_part = []
        for part in parts:
            p, g = pred_np[s] == part, seg_np[s] == part
            union = np.sum(np.logical_or(p, g))
            per_part.append(1.0 if union == 0 else np.sum(np.logical_and(p, g)) / float(union))
        ious.append(np.mean(per_part))
    return ious


# ---- run bookkeeping of the experiment scripts ------------------------------------------------------------
def experiment_details(args, experiment_name):
    """The text experiments/train_modelnet.py:205-209 writes to <logdir>/settings.txt (and prints)."""
    text = experiment_name + '\n--\nSettings:\n--\n'
    for arg in vars(args):
        text += '{}: {}\n'.format(arg, getattr(args, arg))
    return text


def write_settings(args, logdir, experiment_name):
    """<logdir>/settings.txt + <logdir>/checkpoints/ exactly as the reference lays a run out
    (train_modelnet.py:197-211); returns the checkpoint directory."""
    import os
    ckpt = os.path.join(logdir, 'checkpoints')
    os.makedirs(ckpt, exist_ok=True)
    with open(os.path.join(logdir, 'settings.txt'), 'w') as f:
        f.write(experiment_details(args, experiment_name))
    return ckpt


def save_checkpoint(model, path):
    """torch.save(model.state_dict(), path): the reference's checkpoint format (train_modelnet.py:80,82) --
    plain tensors under the reference's parameter names, so either code base loads the other's files."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_checkpoint(model, path, strict=True):
    """model.load_state_dict(torch.load(path)) (train_modelnet.py:84), strict by default."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    return model.load_state_dict(sd, strict=strict)


# ---- multi-vote evaluation of the part-segmentation experiment ----------------------------------------------
@torch.no_grad()
def evaluate_votes(model, loader, num_votes=10, device=None, transform=None, class_choice=None):
    """experiments/test_shapenet.py:71-112: run the test set ``num_votes`` times (the dataset / ``transform``
    re-augments every pass: RandomScale + RandomTranslateGlobal there), SUM the per-point logits of the passes, take
    the arg-max, and report accuracy, mean per-class accuracy and the per-shape part IoU (``calc_shape_IoU``).
    ``transform`` (optional) is applied to every collated batch after the move to ``device`` -- the GPU-side form of
    the augmentation (deltaconv_amd.transforms on a Batch).  Clouds of one loader must have equal sizes, as in the
    reference (it reshapes to [num_graphs, -1, classes])."""
    model.eval()
    acc, true_seg, label_seg = None, [], []
    for vote in range(num_votes):
        preds = []
        for data in loader:
            if device is not None:
                data = data.to(device)
            if transform is not None:
                data = transform(data)
            pred = model(data)
            preds.append(pred.detach().float().cpu().numpy().reshape(data.num_graphs, -1, pred.size(1)))
            if vote == 0:
                true_seg.append(data.y.cpu().numpy().reshape(data.num_graphs, -1))
                if getattr(data, "category", None) is not None:
                    label_seg.append(data.category.max(dim=1)[1].cpu().numpy())
        stacked = np.concatenate(preds, axis=0)
        acc = stacked if acc is None else acc + stacked
    pred_seg = np.argmax(acc, axis=2)
    true_seg = np.concatenate(true_seg, axis=0)
    flat_t, flat_p = true_seg.flatten(), pred_seg.flatten()
    classes = np.unique(flat_t)
    out = dict(pred=pred_seg, true=true_seg, accuracy=float((flat_t == flat_p).mean()),
               balanced_accuracy=float(np.mean([(flat_p[flat_t == c] == c).mean() for c in classes])))
    if label_seg:
        label = np.concatenate(label_seg)
        ious = calc_shape_IoU(pred_seg, true_seg, label, class_choice)
        out.update(label=label, ious=ious, mean_iou=float(np.mean(ious)))
    return out
