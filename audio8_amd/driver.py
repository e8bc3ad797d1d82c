"""Driver-side surface that audio8's pretrain.py / train.py import from the third-party `eight_mile` package
(audio8/pretrain.py:17-21,23; audio8/train.py:15-19), so the unchanged drivers start on a box where mead-baseline is
not installed, plus audio8/utils.py's LR-schedule glue.

  utilities      str2bool, Average, get_num_gpus_multiworker, Offsets, revlut                 (eight_mile.utils)
  runtime        init_distributed, save_checkpoint, find_latest_checkpoint, sequence_mask       (eight_mile.pytorch.layers)
  optimizer      OptimizerManager                                                               (eight_mile.pytorch.optz)
  LR schedules   WarmupLinearScheduler, CosineDecayScheduler, ..., CompositeLRScheduler,
                 create_lr_scheduler, create_lrs                                                (eight_mile.optz, audio8/utils.py:5-15)

eight_mile's source is not in the reference tree (unversioned dependency): what is below follows its published
behaviour as recalled and the call sites in the drivers; where a driver pins a detail it is cited.  `install_aliases()`
registers these under the `eight_mile.*` / `audio8.*` module names the drivers import.
"""
import glob
import logging
import math
import os
import re
import sys
import types

import torch
import torch.distributed as dist

from .ctc import Offsets
from .layers import sequence_mask

logger = logging.getLogger("a8hip")


# ------------------------------------------------------------------ eight_mile.utils
def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise ValueError("Boolean value expected.")


class Average:
    """Running mean with the `name avg` string the drivers log (audio8/pretrain.py:166-167,188-190)."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


def get_num_gpus_multiworker():
    """World size of the job (torch.distributed.run exports WORLD_SIZE)."""
    return int(os.environ.get("WORLD_SIZE", 1))


def revlut(lut):
    return {v: k for k, v in lut.items()}


# ------------------------------------------------------------------ eight_mile.pytorch.layers (driver side)
def init_distributed(local_rank):
    """audio8/pretrain.py:94, train.py:141: one process per GPU, rendezvous from the environment, backend nccl (= RCCL over
    xGMI on ROCm).  Returns (device, local_rank)."""
    if local_rank == -1:
        local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", 0)))
    ngpu = torch.cuda.device_count()
    index = 0 if ngpu == 1 else local_rank % max(1, ngpu)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available():
        torch.cuda.set_device(index)
    if not dist.is_initialized():
        backend = os.environ.get("A8_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, init_method="env://")
    return torch.device("cuda", index) if torch.cuda.is_available() else torch.device("cpu"), local_rank


def checkpoint_for(model_base, count, tick_type="epoch"):
    return f"{model_base}-{tick_type}-{count}"


def save_checkpoint(model, model_base, count, tick_type="epoch", save_npz=False):
    """`{model_base}-{tick_type}-{count}.pth` holding the bare model's state_dict (weights only; DDP's `.module` unwrapped):
    the name audio8/pretrain.py:135-137 parses the step back out of, written at pretrain.py:193 / train.py:394-398."""
    name = checkpoint_for(model_base, count, tick_type) + ".pth"
    if os.path.exists(name):
        return name
    os.makedirs(os.path.dirname(os.path.abspath(name)), exist_ok=True)
    bare = model.module if hasattr(model, "module") else model
    logger.info("Creating checkpoint: %s", name)
    torch.save({k: v.detach().cpu() for k, v in bare.state_dict().items()}, name)
    return name


def find_latest_checkpoint(checkpoint_dir, wildcard="checkpoint"):
    """(path, step) of the checkpoint with the highest step number in the directory (train.py:214-256 resume)."""
    best, best_step = None, -1
    for f in glob.glob(os.path.join(checkpoint_dir, f"{wildcard}*.pth")):
        m = re.search(r"-(\d+)\.pth$", f)
        if m and int(m.group(1)) > best_step:
            best, best_step = f, int(m.group(1))
    if best is None:
        return None, 0
    return best, best_step


# ------------------------------------------------------------------ eight_mile.optz LR schedules + audio8/utils.py
class LearningRateScheduler:
    def __init__(self, **kwargs):
        self.lr = kwargs.get("lr", kwargs.get("eta", 1.0))


class ConstantScheduler(LearningRateScheduler):
    def __call__(self, global_step):
        return self.lr


class WarmupLinearScheduler(LearningRateScheduler):
    """lr * min(1, step / warmup_steps)."""

    def __init__(self, warmup_steps=16000, **kwargs):
        super().__init__(**kwargs)
        self.warmup_steps = warmup_steps

    def __call__(self, global_step):
        return self.lr * min(1.0, global_step / float(max(1, self.warmup_steps)))


WarmupLinearSchedulerPyTorch = WarmupLinearScheduler


class CosineDecayScheduler(LearningRateScheduler):
    """lr * ((1 - alpha) * 0.5 * (1 + cos(pi * min(step, decay_steps) / decay_steps)) + alpha)."""

    def __init__(self, decay_steps=1000, alpha=0.0, **kwargs):
        super().__init__(**kwargs)
        self.decay_steps, self.alpha = decay_steps, alpha

    def __call__(self, global_step):
        step = min(global_step, self.decay_steps)
        cosine = 0.5 * (1.0 + math.cos(math.pi * step / float(max(1, self.decay_steps))))
        return self.lr * ((1.0 - self.alpha) * cosine + self.alpha)


class ExponentialDecayScheduler(LearningRateScheduler):
    def __init__(self, decay_steps=1000, decay_rate=0.5, staircase=False, **kwargs):
        super().__init__(**kwargs)
        self.decay_steps, self.decay_rate, self.staircase = decay_steps, decay_rate, staircase

    def __call__(self, global_step):
        e = global_step / float(self.decay_steps)
        return self.lr * self.decay_rate ** (math.floor(e) if self.staircase else e)


class CompositeLRScheduler(LearningRateScheduler):
    """warm-up, then `plateau_steps` at the base rate, then the decay schedule counted from its own step 0."""

    def __init__(self, warm=None, rest=None, plateau_steps=0, **kwargs):
        super().__init__(**kwargs)
        self.warm, self.rest, self.plateau_steps = warm, rest, plateau_steps

    def __call__(self, global_step):
        w = self.warm.warmup_steps
        if global_step < w:
            return self.warm(global_step)
        if global_step < w + self.plateau_steps:
            return self.lr
        return self.rest(global_step - w - self.plateau_steps)


_SCHEDULERS = {"cosine": CosineDecayScheduler, "exponential": ExponentialDecayScheduler, "constant": ConstantScheduler,
               "default": ConstantScheduler, "warmup_linear": WarmupLinearScheduler}


def create_lr_scheduler(lr_scheduler_type="default", **kwargs):
    if lr_scheduler_type not in _SCHEDULERS:
        raise ValueError(f"a8hip: unknown lr_scheduler_type {lr_scheduler_type!r} (have {sorted(_SCHEDULERS)})")
    return _SCHEDULERS[lr_scheduler_type](**kwargs)


def get_lr_decay(lr, decay_steps, sched_type="cosine", alpha=0.0):
    """audio8/utils.py:5-8."""
    return create_lr_scheduler(lr_scheduler_type=sched_type, lr=lr, decay_steps=decay_steps, alpha=alpha)


def create_lrs(lr, train_steps, sched_type="cosine", alpha=0.0, warmup_steps=10000, plateau_steps=0, **kwargs):
    """audio8/utils.py:11-15: linear warm-up -> plateau -> cosine (default) decay over train_steps."""
    return CompositeLRScheduler(WarmupLinearScheduler(warmup_steps, lr=lr), get_lr_decay(lr, train_steps, sched_type, alpha), plateau_steps, lr=lr)


# ------------------------------------------------------------------ eight_mile.pytorch.optz.OptimizerManager
class OptimizerManager:
    """`OptimizerManager(model, global_step, optim=, lr=, lr_function=, weight_decay=)` with the members the drivers use:
    zero_grad(), step(), scale_grads(s), current_lr, global_step (audio8/pretrain.py:146-148,171-190; train.py:254-256,
    292-346).  On the GPU it is a front for audio8_amd.trainer.Stepper (flat f32 state, fused clip + AdamW, bucketed RCCL
    exchange); gradient clipping stays where the driver put it (torch.nn.utils.clip_grad_norm_ over the flat views) unless
    `clip` is given, which fuses it into the update.

    The learning rate of a step is lr_function(global_step) with global_step as it stands before the step increments it
    (the first step of a fresh run uses lr_function(0), a run resumed at step N continues with lr_function(N))."""

    def __init__(self, model, global_step=0, optim="adamw", lr=2.0e-4, lr_function=None, weight_decay=0.0, clip=0.0, **kwargs):
        from .trainer import Stepper
        if optim not in ("adam", "adamw"):
            raise NotImplementedError("a8hip OptimizerManager: the fused update is Adam / AdamW (audio8's drivers default to adamw)")
        bare = model.module if hasattr(model, "module") else model
        self.lr_function = lr_function if lr_function is not None else ConstantScheduler(lr=lr)
        self.stepper = Stepper(bare, lr=lr, weight_decay=weight_decay if optim == "adamw" else 0.0, clip=clip,
                               lr_function=self.lr_function, global_step=global_step,
                               betas=(kwargs.get("beta1", 0.9), kwargs.get("beta2", 0.999)), eps=kwargs.get("epsilon", 1e-8))

    @property
    def global_step(self):
        return self.stepper.global_step

    @property
    def current_lr(self):
        return self.stepper.current_lr

    def zero_grad(self):
        self.stepper.zero_grad()

    def scale_grads(self, s):
        """train.py:323 `optimizer.scale_grads(num_gpus / batch_size)`; `s` may be a device tensor.  The driver calls it between
        backward() and step(), where DDP has already averaged the gradients: Stepper.sync_grads() first completes the exchange
        (weight-gradient stream joined, every bucket reduced) and turns the flat buffer into that mean.  (The native form is
        `Stepper.step(batch_size=...)`, which needs no pass over the gradients.)"""
        self.stepper.sync_grads()
        self.stepper.state.flat_g.mul_(s if torch.is_tensor(s) else float(s))

    def step(self):
        self.stepper.step()


class PassThroughDDP(torch.nn.Module):
    """Stand-in for torch's DistributedDataParallel when the gradient exchange is the Stepper's (bucketed RCCL all-reduce
    issued by the backward kernels' own gradient-ready marks; torch's DDP reducer never sees gradients that are written
    straight into the flat buffer).  Same surface as the drivers use: `.module`, call-through, `no_sync()`
    (audio8/pretrain.py:158, train.py:266-270,301)."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, **kwargs):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def no_sync(self):
        from . import trainer
        if trainer.ACTIVE_STEPPER is None:
            raise RuntimeError("a8hip: no_sync() needs a Stepper / OptimizerManager for this model")
        return trainer.ACTIVE_STEPPER.no_sync()


_TORCH_CLIP_GRAD_NORM = torch.nn.utils.clip_grad_norm_


def clip_grad_norm_(parameters, max_norm, *args, **kwargs):
    """`torch.nn.utils.clip_grad_norm_` for the drivers (audio8/pretrain.py:182, train.py:324).  Under DDP every `.grad` is final
    and averaged when backward() returns; here gradients may still be in flight on the weight-gradient stream and in the bucket
    all-reduces, and hold the SUM over ranks: the active Stepper completes the exchange and forms the mean first
    (Stepper.sync_grads()), then torch's own function runs on the flat views."""
    from . import trainer
    if trainer.ACTIVE_STEPPER is not None:
        trainer.ACTIVE_STEPPER.sync_grads()
    return _TORCH_CLIP_GRAD_NORM(parameters, max_norm, *args, **kwargs)


# ------------------------------------------------------------------ module aliases for the unchanged drivers
def install_aliases(audio8_too=True, patch_ddp=True):
    """Register this package under the module names audio8's drivers import, ahead of (absent) eight_mile:
        eight_mile.utils, eight_mile.pytorch.layers, eight_mile.pytorch.optz, eight_mile.optz, eight_mile.pytorch.serialize,
        audio8.utils (create_lrs) and -- with audio8_too -- audio8.wav2vec2 / audio8.ctc.
    Existing real modules are never replaced.  patch_ddp: `torch.nn.parallel.DistributedDataParallel` becomes PassThroughDDP,
    because the Stepper inside OptimizerManager already exchanges the gradients (an all-reduce by DDP on top would average
    twice, and DDP's reducer cannot see gradients the kernels accumulate in place) and `torch.nn.utils.clip_grad_norm_` first
    completes that exchange (clip_grad_norm_ above)."""
    from . import ctc as _ctc, layers as _layers, wav2vec2 as _w2v
    me = sys.modules[__name__]

    def mod(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    pub = {k: getattr(me, k) for k in ("str2bool", "Average", "get_num_gpus_multiworker", "Offsets", "revlut")}
    mod("eight_mile")
    mod("eight_mile.utils", **pub)
    mod("eight_mile.pytorch")
    lay = {k: getattr(_layers, k) for k in dir(_layers) if not k.startswith("_")}
    lay.update(save_checkpoint=save_checkpoint, init_distributed=init_distributed, find_latest_checkpoint=find_latest_checkpoint,
               sequence_mask=sequence_mask, checkpoint_for=checkpoint_for)
    mod("eight_mile.pytorch.layers", **lay)
    sched = {k: getattr(me, k) for k in ("LearningRateScheduler", "ConstantScheduler", "WarmupLinearScheduler", "WarmupLinearSchedulerPyTorch",
                                         "CosineDecayScheduler", "ExponentialDecayScheduler", "CompositeLRScheduler", "create_lr_scheduler")}
    mod("eight_mile.optz", **sched)
    mod("eight_mile.pytorch.optz", OptimizerManager=OptimizerManager, __all__=["OptimizerManager"] + sorted(sched), **sched)
    mod("eight_mile.pytorch.serialize", convert_transformers_keys=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("out of scope")),
        load_tlm_npz=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("out of scope")))
    if patch_ddp:
        torch.nn.parallel.DistributedDataParallel = PassThroughDDP
        torch.nn.utils.clip_grad_norm_ = clip_grad_norm_        # the drivers look it up at call time
    if audio8_too:
        mod("audio8")
        mod("audio8.utils", create_lrs=create_lrs, get_lr_decay=get_lr_decay)
        sys.modules.setdefault("audio8.wav2vec2", _w2v)
        sys.modules.setdefault("audio8.ctc", _ctc)
