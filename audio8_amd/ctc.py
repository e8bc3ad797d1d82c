"""MI355X-native mirror of audio8/ctc.py: CTCLoss (audio8/ctc.py:186-206) on the wave-per-utterance
HIP kernel, and the greedy-decode metrics (audio8/ctc.py:139-183, 63-72) with an own Levenshtein
(the reference imports the third-party `editdistance`).  The prefix-beam/LM decoder
(audio8/ctc.py:11-60) wraps the third-party `ctcdecode` and is out of scope."""
import numpy as np
import torch
import torch.nn

from . import kernels as K
from . import ops


class Offsets:
    """eight_mile.utils.Offsets as audio8/train.py:20-27 configures it."""
    GO, PAD, EOS, UNK = 0, 1, 2, 3
    VALUES = ['<s>', '<pad>', '</s>', '<unk>']


def postproc_letters(sentence):
    return ''.join(sentence).replace(" ", "").replace("|", " ").strip()


def postproc_bpe(sentence):
    return ' '.join(sentence).replace("@@ ", "").strip()


def edit_distance(a, b) -> int:
    """Levenshtein distance between two sequences (== editdistance.eval)."""
    a, b = list(a), list(b)
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        prev = cur
    return prev[-1]


def _greedy_units(lprobs_t, input_lengths):
    """Per utterance: argmax -> unique_consecutive -> drop blanks (audio8/ctc.py:156-159).  Device tensors go through the
    a8_ctc_greedy_decode kernel (one D2H copy of the token table); host tensors are decoded on the host like the reference."""
    BLANK_IDX = Offsets.GO
    if lprobs_t.is_cuda:
        from . import kernels as K
        in_len = (input_lengths if torch.is_tensor(input_lengths) else torch.as_tensor(input_lengths)).to(device=lprobs_t.device, dtype=torch.int64)
        in_len = in_len.clamp(max=lprobs_t.shape[1]).to(torch.int32).contiguous()
        tokens, lens = K.ctc_greedy_decode(lprobs_t, in_len, BLANK_IDX)
        tokens, lens = tokens.cpu().numpy(), lens.cpu().numpy()
        return [tokens[b, :lens[b]].tolist() for b in range(tokens.shape[0])]
    out = []
    for lp, inp_l in zip(lprobs_t, input_lengths):
        toks = lp[:inp_l].argmax(dim=-1).unique_consecutive()
        out.append(toks[toks != BLANK_IDX].tolist())
    return out


def ctc_metrics(lprobs_t, target, input_lengths, index2vocab, postproc_fn=postproc_letters):
    """Greedy CTC decode + CER/WER counts.  Keeps the reference's quirk of slicing frames by the
    *sample* length (audio8/ctc.py:156 with audio8/train.py:47)."""
    c_err = c_len = w_errs = w_len = wv_errs = 0
    with torch.no_grad():
        preds = _greedy_units(lprobs_t, input_lengths)
        target = target.cpu() if torch.is_tensor(target) and target.is_cuda else target
        for pred_units_arr, t in zip(preds, target):
            p = (t != Offsets.PAD) & (t != Offsets.EOS)
            targ = t[p]
            targ_units_arr = targ.tolist()
            c_err += edit_distance(pred_units_arr, targ_units_arr)
            c_len += len(targ_units_arr)
            targ_words = postproc_fn([index2vocab[x] for x in targ_units_arr]).split()
            pred_words_raw = postproc_fn([index2vocab[x] for x in pred_units_arr]).split()
            dist = edit_distance(pred_words_raw, targ_words)
            w_errs += dist
            wv_errs += dist
            w_len += len(targ_words)
    return {"wv_errors": wv_errs, "w_errors": w_errs, "w_total": w_len, "c_errors": c_err, "c_total": c_len}


class CTCLoss(torch.nn.Module):
    """forward(log_prob [T,B,V], input_lengths [B], targets int [B,S] padded with PAD/EOS, target_lengths [B])."""

    def __init__(self, zero_infinity=True, reduction_type="sum"):
        super().__init__()
        if reduction_type not in ("sum", "mean", "none"):
            raise ValueError(f"unknown reduction {reduction_type}")
        self.zero_infinity = zero_infinity
        self.reduction_type = reduction_type
        self._pending_status = None

    def _check_deferred(self):
        """The device-side target packing reports rows whose non-PAD/EOS count differs from target_lengths (the reference's
        masked_select + F.ctc_loss would fail on them): raised at the NEXT call, when the flag has long arrived."""
        st, self._pending_status = self._pending_status, None
        if st is not None and int(st[0].item()) != 0:
            raise RuntimeError("CTCLoss: %d utterance(s) of the previous batch had target_lengths that did not match their non-PAD/EOS "
                               "targets (their labels were cut / zero-filled)" % int(st[0].item()))

    def forward(self, log_prob, input_lengths, targets, target_lengths):
        dev = log_prob.device
        in_len = input_lengths.to(device=dev, dtype=torch.int32) if torch.is_tensor(input_lengths) else ops.host_i32(input_lengths, dev)
        if torch.is_tensor(targets) and targets.is_cuda and targets.dim() == 2:
            # device targets (audio8/train.py:35): PAD/EOS stripping and the offsets stay on the device -- no host round trip
            # (the reference's masked_select is one).  The label bound is the padded width of `targets`.
            if torch.is_tensor(target_lengths):
                tl_dev = target_lengths.to(device=dev, dtype=torch.int32) if target_lengths.is_cuda else ops.host_i32(target_lengths.numpy(), dev)
            else:
                tl_dev = ops.host_i32(np.asarray(target_lengths), dev)
            self._check_deferred()                       # last call's consistency flag (read one call late: no sync on this one)
            flat, off, status = K.ctc_pack_targets(targets.to(torch.int64).contiguous(), tl_dev.contiguous(), Offsets.PAD, Offsets.EOS)
            self._pending_status = status
            # label bound of the alpha/beta kernel (<= 511 labels): the true maximum when target_lengths lives on the host (the
            # drivers' batches, audio8/train.py:30-39), else the padded width of `targets` -- and only if that is too wide, one sync
            if not (torch.is_tensor(target_lengths) and target_lengths.is_cuda):
                tl_host = np.asarray(target_lengths.detach().numpy() if torch.is_tensor(target_lengths) else target_lengths)
                smax = int(max(1, tl_host.max() if tl_host.size else 1))
            else:
                smax = max(1, int(targets.shape[1]))
                if smax > 511:
                    smax = int(max(1, int(target_lengths.max().item())))
            tl = None
        else:
            t = np.asarray(targets.detach().cpu().numpy() if torch.is_tensor(targets) else targets)
            flat_np = t[(t != Offsets.PAD) & (t != Offsets.EOS)]                  # audio8/ctc.py:193-194
            tl = np.asarray(target_lengths.detach().cpu().numpy() if torch.is_tensor(target_lengths) else target_lengths).astype(np.int64)
            if int(tl.sum()) != flat_np.size:
                raise RuntimeError("CTCLoss: sum(target_lengths) does not match the number of non-PAD/EOS targets")
            flat, off = ops.host_i32(flat_np, dev), ops.host_i32(np.concatenate([[0], np.cumsum(tl)]), dev)
            smax = int(max(1, tl.max() if tl.size else 1))
        nll = ops.CTCFn.apply(log_prob, flat, off, in_len.contiguous(), smax, Offsets.GO, self.zero_infinity)
        if self.reduction_type == "sum":
            return nll.sum()
        if self.reduction_type == "mean":                                     # ATen: mean over batch of nll / clamp(target_len, 1)
            if tl is None:
                tl = np.asarray(target_lengths.detach().cpu().numpy() if torch.is_tensor(target_lengths) else target_lengths).astype(np.int64)
            denom = torch.from_numpy(np.maximum(tl, 1)).to(device=dev, dtype=nll.dtype)
            return (nll / denom).mean()
        return nll
