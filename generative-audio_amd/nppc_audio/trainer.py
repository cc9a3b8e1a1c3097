"""NPPC speech-enhancement trainer on the MI355X kernels: same surface as nppc_audio/trainer.py:20-371.

  NPPCAudioTrainerConfig / OptimizerConfig           trainer.py:20-37
  NPPCAudioTrainer(config).train / base_step / save_checkpoint / _calculate_final_objective /
  _get_true_and_pred_crm                              trainer.py:40-371

What differs, deliberately:
  * every tensor op of the step that touches a map, an activation or a weight is a HIP kernel of libnppc_hip.so (what is
    left to torch is scalar assembly: the mean + lambda * mean of two small loss vectors);
  * the frozen restorer and the noisy STFT run once per step (the reference repeats them 2x / 3x with identical
    results);
  * data parallelism (the reference has none): one process per GPU, contiguous equal shards of the minibatch,
    ONE gradient all-reduce (mean) of the flat direction-net gradient over RCCL, optimizer replicated;
  * `optimizer_configuration.type == "Adam"` is served by `HipAdam`, a torch.optim.Optimizer with torch's Adam
    semantics / state_dict layout whose step is one fused kernel over the flat parameter buffer.
"""
import json
import os
from datetime import datetime

import pydantic
import torch
import torch.nn as nn
import torch.optim as optim

from . import _hip as H
from . import ops
from . import ops_lstm
from .data import AudioDataset, DataConfig, DataLoaderConfig, DeviceMixLoader
from .dp import FlatGradientReducer, mean_reduce_parameter_grads
from .nppc_model import NPPCModel, NPPCModelConfig
from .pc_ops import NPPCLoss, second_moment_weight


class OptimizerConfig(pydantic.BaseModel):
    type: str
    args: dict


class NPPCAudioTrainerConfig(pydantic.BaseModel):
    """Configuration for NPPCAudio trainer"""
    nppc_model_configuration: NPPCModelConfig
    data_configuration: DataConfig
    data_loader_configuration: DataLoaderConfig
    optimizer_configuration: OptimizerConfig
    learning_rate: float = 1e-4
    device: str = "cuda"
    save_interval: int = 10
    log_interval: int = 100
    second_moment_loss_lambda: float = 1.0
    second_moment_loss_grace: int = 500


class LoopLoader(object):
    """nppc/auxil.py:124-147: re-iterate a DataLoader until n_steps batches were produced."""

    def __init__(self, dataloader, n_steps=None, n_epochs=None):
        if n_epochs is not None:
            n_steps = n_epochs * len(dataloader) if n_steps is None else min(n_steps, n_epochs * len(dataloader))
        if n_steps is None:
            raise Exception('Either "n_steps" or "n_epochs" must be provided.')
        self.dataloader = dataloader
        self._n = n_steps

    def __len__(self):
        return self._n

    def __iter__(self):
        i = 0
        while i < len(self):
            for x in self.dataloader:
                if i == len(self):
                    break
                yield x
                i += 1


def get_true_and_pred_crm(model, clean_waveform, noisy_waveform, num_groups_in_drop_band):
    """trainer.py:344-371 -> (gt_crm, pred_crm), both compressed cIRM [B,2,F',T] in drop-band order."""
    st = model.config.stft_configuration
    f = model._front(noisy_waveform)                      # noisy STFT + restorer: shared with NPPCModel.forward
    _, c_re, c_im = ops.stft(clean_waveform, st.nfft, st.hop_length, want_mag=False)
    gt = ops.cirm_build_compress(f["re"], f["im"], c_re, c_im, num_groups_in_drop_band)
    pred = ops.drop_band(model.get_pred_crm(noisy_waveform), num_groups_in_drop_band)
    return gt, pred


def nppc_base_step(model, batch, step, grace, lam_cfg):
    """trainer.py:234-317 as a free function: (reconst_err [B], objective [], log dict)."""
    noisy_waveform, clean_waveform = batch
    w_mat = model(noisy_waveform)                          # [B, n_dirs, 2, F', T]
    groups = model.config.audio_pc_wrapper_configuration.multi_direction_configuration.num_groups_in_drop_band
    gt_crm, pred_crm = get_true_and_pred_crm(model, clean_waveform, noisy_waveform, groups)
    lam = second_moment_weight(step, grace, lam_cfg)
    reconst_err, objective, err_norm, pr, pi, pm, w_norms, sm = NPPCLoss.apply(w_mat, gt_crm, pred_crm, lam)
    log = {
        'noisy_complex': noisy_waveform,
        'clean_complex': clean_waveform,
        'pred_crm': pred_crm.detach(),
        'w_mat': w_mat.detach(),
        'err_norm': err_norm.detach(),
        'err_proj': torch.complex(pr, pi).detach(),
        'err_proj_mag': pm.detach(),
        'w_norms': w_norms.detach(),
        'reconst_err': reconst_err.detach(),
        'second_moment_mse': sm.detach(),
        'objective': objective.detach(),
    }
    return reconst_err, objective, log


class HipAdam(optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad / maximize) with the update done by nppc_adam_step.

    Parameters whose storage and gradient are views of an engine's flat buffers are updated with ONE kernel
    per flat buffer; any other parameter with a gradient gets the same kernel on its own storage.  Parameters
    without a gradient are skipped, exactly like torch (the frozen restorer's weights are handed to the
    optimizer by the reference too, trainer.py:66-69).  state_dict() has torch.optim.Adam's layout.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **unsupported):
        for k in ("amsgrad", "maximize", "capturable", "differentiable"):
            if unsupported.pop(k, False):
                raise NotImplementedError(f"HipAdam does not implement {k}=True")
        unsupported.pop("foreach", None), unsupported.pop("fused", None)
        if unsupported:
            raise TypeError(f"unexpected Adam arguments: {sorted(unsupported)}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        H.require_gpu()
        s = H.stream()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                H.call("nppc_adam_step", p.data, g, st["exp_avg"], st["exp_avg_sq"], p.numel(), float(group["lr"]),
                       float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), int(st["step"]),
                       float(self.grad_scale), s)
                torch.autograd.graph.increment_version(p)       # written through a raw pointer: tell the engines
        return loss


PREPACK_AFTER_UPDATE = os.environ.get("NPPC_PREPACK", "1") != "0"     # A/B switch (tools/diag)


class FlatAdamStepper:
    """Fast path used by NPPCAudioTrainer.train: one Adam kernel over the direction net's flat parameter /
    gradient buffers, sharing state tensors with a HipAdam instance so checkpoints stay interchangeable."""

    def __init__(self, opt: HipAdam, engine):
        self.opt, self.eng = opt, engine
        fp = engine.fp
        n = fp.flat.numel()
        self.m = torch.zeros(n, dtype=torch.float32, device=fp.flat.device)
        self.v = torch.zeros_like(self.m)
        self.t = 0
        group_of = {}
        for gi, g in enumerate(opt.param_groups):
            for p in g["params"]:
                group_of[id(p)] = g
        self.group = None
        self._step_t = torch.tensor(0.0)
        for name, p in fp.named:
            g = group_of.get(id(p))
            if g is None:
                raise RuntimeError(f"{name} is not in the optimizer")
            if self.group is None:
                self.group = g
            elif g is not self.group:
                raise RuntimeError("flat Adam needs the direction net in ONE param group")
            o, shp = fp.off[name]
            st = opt.state[p]
            k = p.numel()
            if st:                                   # adopt existing state (resumed checkpoint)
                self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.t = int(st["step"])
            st["step"] = self._step_t               # ONE shared host scalar: 340 per-parameter fills per step took 0.7 ms of host time
            st["exp_avg"] = self.m[o:o + k].view(shp)
            st["exp_avg_sq"] = self.v[o:o + k].view(shp)
        self._step_t.fill_(float(self.t))

    def step(self, gflat, grad_scale=1.0, clip=None, poison=None):
        """clip = (device double holding sum(g^2), max_norm): clip_grad_norm_ folded into the same kernel.
        The plain update is guarded by the sticky hand-off time-out counters of the cooperative LSTM kernels: when one is
        non-zero the kernel leaves parameters and moments untouched and writes NaN into `poison` (the step's objective)."""
        self.t += 1
        g = self.group
        b1, b2 = g["betas"]
        if clip is None:
            guards, ng = ops_lstm.timeout_guards()
            H.call("nppc_adam_step_guarded", self.eng.fp.flat, gflat, self.m, self.v, gflat.numel(), float(g["lr"]), float(b1),
                   float(b2), float(g["eps"]), float(g["weight_decay"]), self.t, float(grad_scale), guards, ng, poison, H.stream())
        else:
            H.call("nppc_adam_step_clip", self.eng.fp.flat, gflat, self.m, self.v, gflat.numel(), float(g["lr"]),
                   float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), self.t, float(grad_scale), clip[0],
                   float(clip[1]), H.stream())
        torch.autograd.graph.increment_version(self.eng.fp.flat)   # packed copies of the weights are stale now
        self._step_t.fill_(float(self.t))


class NPPCAudioTrainer(nn.Module):
    def __init__(self, config: NPPCAudioTrainerConfig, dataset=None):
        super().__init__()
        self.config = config
        self.nppc_model = NPPCModel(self.config.nppc_model_configuration)
        self.device = self.config.device
        if dataset is None:
            dataset = AudioDataset(config.data_configuration.dataset)
        print(f"Total sample pairs in dataset: {len(dataset)}")
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        lc = config.data_loader_configuration
        bs = lc.batch_size
        batch_sampler = None
        if self.world > 1:
            # contiguous equal shards of every global minibatch: rank r takes samples
            # [r*bs/W, (r+1)*bs/W) of each batch, so `i_global mod G == i_local mod G` for even local batches
            assert bs % self.world == 0, "global batch must divide evenly over the ranks"
            batch_sampler = ShardedBatchSampler(len(dataset), bs, self.world, self.rank, lc.shuffle)
        if isinstance(dataset, AudioDataset) and str(self.device).startswith("cuda"):
            # wav-folder dataset on a HIP device: crops are cut on the host, the minibatch is mixed by ONE kernel launch
            if batch_sampler is None:
                base = (torch.utils.data.RandomSampler if lc.shuffle else torch.utils.data.SequentialSampler)(dataset)
                batch_sampler = torch.utils.data.BatchSampler(base, bs, drop_last=False)
            self.dataloader = DeviceMixLoader(dataset, batch_sampler, device=self.device, pin_memory=lc.pin_memory)
        elif batch_sampler is not None:
            self.dataloader = torch.utils.data.DataLoader(dataset, batch_sampler=batch_sampler, num_workers=lc.num_workers,
                                                          pin_memory=lc.pin_memory)
        else:
            self.dataloader = torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=lc.shuffle,
                                                          num_workers=lc.num_workers, pin_memory=lc.pin_memory)
        self.step = 0
        # the frozen restorer never needs gradient buffers; the reference leaves requires_grad on and relies on
        # no_grad (nppc_model.py:94) -- same effect, and its parameters are still handed to the optimizer below
        okind = config.optimizer_configuration.type
        if okind == "Adam":
            self.optimizer = HipAdam(self.nppc_model.parameters(), **config.optimizer_configuration.args)
        else:
            self.optimizer = getattr(optim, okind)(self.nppc_model.parameters(), **config.optimizer_configuration.args)
        self._flat_adam = None
        # Data parallel: ONE all-reduce of the flat gradient on the main stream behind backward is the default; the bucketed
        # exchange that starts inside backward (dp.py) and the parked update (pipeline_update) are opt-in
        # (NPPC_DP_OVERLAP=1 / NPPC_DP_PIPELINE=1) until a multi-rank RCCL run of the bf16 step with the cooperative LSTM
        # kernels is on record: RCCL's persistent kernels beside co-residency-dependent cooperative kernels is exactly what
        # the one-GPU boxes of this pool cannot exercise.
        self._reducer = FlatGradientReducer(overlap=os.environ.get("NPPC_DP_OVERLAP", "0") == "1")
        # pipeline_update: the tail of a step (side-stream weight gradients, the rest of the gradient exchange, Adam) is not
        # waited for at the end of train_step but parked in front of the NEXT step's restorer LSTM launch, so it overlaps
        # the next minibatch's STFTs and the frozen restorer's full-band front (which do not read the trained weights).
        # The update is applied before the direction net runs again: results are identical; parameters are current after
        # flush() (train() checks the hand-off time-out counter at every log interval and flushes at the end,
        # save_checkpoint() always flushes).
        self.pipeline_update = False
        self._seed = None
        self._pending = None
        self._poison = None

    # ---------------------------------------------------------------------------------- reference API
    def base_step(self, batch):
        return nppc_base_step(self.nppc_model, batch, self.step, self.config.second_moment_loss_grace,
                              self.config.second_moment_loss_lambda)

    def _calculate_final_objective(self, reconst_err, second_moment_mse):
        lam = second_moment_weight(self.step, self.config.second_moment_loss_grace, self.config.second_moment_loss_lambda)
        return reconst_err.mean() + lam * second_moment_mse.mean()

    def _get_true_and_pred_crm(self, clean_waveform, model, noisy_waveform, num_groups_in_drop_band):
        return get_true_and_pred_crm(model, clean_waveform, noisy_waveform, num_groups_in_drop_band)

    # ---------------------------------------------------------------------------------- one optimisation step
    def train_step(self, batch):
        """forward + loss + backward (+ gradient all-reduce over RCCL) + Adam; returns (reconst_err, objective, log).

        With HipAdam (the reference's yaml optimizer) the direction net's gradient never leaves its flat buffer:
        backward writes it, ONE all-reduce sums it across ranks, ONE Adam launch applies it (1/world folded into the
        kernel's gradient scale).  Any other torch optimizer goes through .grad like the reference loop."""
        net = self.nppc_model.audio_pc_wrapper.net
        fast = isinstance(self.optimizer, HipAdam)
        net.flat_grad_only = fast
        eng = net.engine()
        # buckets of the flat gradient start their all-reduce from inside backward as soon as they are final (dp.py)
        eng.grad_range_hook = self._reducer.range_ready if (fast and self.world > 1) else None
        defer = fast and self.pipeline_update
        eng.defer_join = defer
        if self._pending is not None:
            # the previous step's update goes out right before the restorer's LSTM launch of THIS step
            self.nppc_model.pretrained_restoration_model.engine().pre_lstm_hook = self.flush
        ok = False
        try:
            reconst_err, objective, log = self.base_step(batch)
            self.flush()                                     # (no-op when the hook already ran)
            self.optimizer.zero_grad()
            if self._seed is None or self._seed.device != objective.device:
                self._seed = torch.ones((), dtype=objective.dtype, device=objective.device)   # d objective / d objective, built once
            objective.backward(self._seed)                   # (backward() alone fills a fresh ones tensor every step)
            ok = True
        finally:
            net.flat_grad_only = False
            eng.grad_range_hook = None
            eng.defer_join = False
            if not ok:
                self._reducer.reset()                        # an exception inside backward must not leave buckets behind
        if fast:
            self._pending = eng
            obj_d = objective.detach()
            self._poison = obj_d if (obj_d.dtype == torch.float32 and obj_d.is_contiguous() and obj_d.is_cuda) else None
            if not defer:
                self._apply_pending(prepack=PREPACK_AFTER_UPDATE)
        else:
            mean_reduce_parameter_grads(net.parameters())
            self.optimizer.step()
        self.step += 1
        return reconst_err, objective, log

    def flush(self):
        """apply a parked update now (join the weight-gradient stream, finish the gradient exchange, one Adam launch) and join
        whatever the last step left on the direction net's side stream: parameters AND streams are current afterwards"""
        self._apply_pending(prepack=False)
        eng = getattr(self.nppc_model.audio_pc_wrapper.net, "_engine", None)     # (no engine yet: nothing was launched)
        if eng is not None and eng.join_pending:
            reng = getattr(self.nppc_model.pretrained_restoration_model, "_engine", None)
            if reng is not None:
                reng.pre_lstm_hook = None
            eng.join_side()

    def _apply_pending(self, prepack):
        eng, self._pending = self._pending, None
        if eng is None:
            return
        reng = self.nppc_model.pretrained_restoration_model.engine()
        reng.pre_lstm_hook = None
        eng.join_side()
        gflat = eng.fp.grad
        scale = self._reducer.finish(gflat)                  # sum over ranks; the mean's 1/W goes into Adam's grad scale
        if self._flat_adam is None or self._flat_adam.eng is not eng:
            self._flat_adam = FlatAdamStepper(self.optimizer, eng)
        self._flat_adam.step(gflat, scale, poison=self._poison)
        self._poison = None
        if prepack and eng.side_stream() is not None:
            # the bf16 / packed copies of the updated weights (0.2 ms of small launches at the top of the direction net's next
            # forward, i.e. on the critical path between the two forward LSTMs) go out NOW on the side stream: they overlap
            # the next minibatch's STFT and the frozen restorer's front, and are joined right before the restorer's LSTM
            # launch (the cooperative kernels want the chip to themselves: FSNEngine._unjoined)
            eng.prepack_on_side()
            reng.pre_lstm_hook = eng.join_side

    def train(self, n_steps=None, n_epochs=None, checkpoint_dir="checkpoints", log_every=None):
        os.makedirs(checkpoint_dir, exist_ok=True)
        loop_loader = LoopLoader(dataloader=self.dataloader, n_steps=n_steps, n_epochs=n_epochs)
        log_every = log_every or self.config.log_interval
        log_dict = None
        if len(loop_loader) == 0:
            raise ValueError("the data loader yields no minibatch (fewer samples than one global batch?)")
        # pipeline_update would hide the tail of the gradient exchange behind the next step's front: opt-in (see __init__)
        self.pipeline_update = self.world > 1 and os.environ.get("NPPC_DP_PIPELINE", "0") == "1"
        try:
            self._train_loop(loop_loader, log_every)
        finally:
            self.pipeline_update = False
        log_dict = self._last_log
        timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        if self.rank == 0 and log_dict is not None:
            self._get_and_save_metrics(checkpoint_dir, log_dict, n_epochs, n_steps, timestamp)
            self.save_checkpoint(os.path.join(checkpoint_dir, f"checkpoint_final_{timestamp}.pt"))

    def _check_timeouts(self, where):
        """the sticky hand-off time-out counter, agreed on by ALL ranks (a rank that raised alone would leave its peers
        hanging in the next all-reduce): every rank raises when any rank counted one"""
        n = ops_lstm.coop_timeouts()
        if self.world > 1:
            import torch.distributed as dist
            t = torch.tensor([n], dtype=torch.int64, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if int(t.item()) and not n:
                raise RuntimeError(f"cooperative LSTM hand-off time-out(s) on another rank at {where}: results since the "
                                   "last check are invalid")
        if n:
            ops_lstm.check_coop_timeouts(where)

    def _train_loop(self, loop_loader, log_every):
        log_dict = None
        self._last_log = None
        for it, batch in enumerate(loop_loader):
            if isinstance(batch, (tuple, list)):
                batch = tuple(x.to(self.device, non_blocking=True) for x in batch)
            else:
                batch = batch.to(self.device)
            reconst_err, objective, log_dict = self.train_step(batch)
            if it % log_every == 0 or it + 1 == len(loop_loader):
                # host syncs only every log_every steps (the reference does three .item() per step, trainer.py:107-113);
                # the same sync point checks the sticky hand-off time-out counter of the cooperative LSTM kernels, on
                # every rank: after a time-out the kernels continue with wrong numbers
                self._check_timeouts(f"step {self.step}")
            if self.rank == 0 and (it % log_every == 0 or it + 1 == len(loop_loader)):
                print(f'step {self.step}: Objective: {objective.item():.4f} | '
                      f'Second Moment MSE: {log_dict["second_moment_mse"].mean().item():.4f} | '
                      f'Reconstract Error: {reconst_err.mean().item():.4f}')
            self._last_log = log_dict
        self.flush()

    def _get_and_save_metrics(self, checkpoint_dir, log_dict, n_epochs, n_steps, timestamp):
        ds = self.config.data_configuration.dataset
        final_metrics = {
            'timestamp': timestamp,
            'total_steps': self.step,
            'final_loss': log_dict['objective'].item(),
            'final_second_moment_mse': log_dict['second_moment_mse'].mean().item(),
            'training_config': {
                'n_steps': n_steps, 'n_epochs': n_epochs, 'learning_rate': self.config.learning_rate,
                'device': self.config.device, 'snr_range': list(ds.snr_range),
                'sub_sample_length_seconds': ds.sub_sample_length_seconds,
                'batch_size': self.config.data_loader_configuration.batch_size,
            },
        }
        with open(os.path.join(checkpoint_dir, f"metrics_final_{timestamp}.json"), 'w') as f:
            json.dump(final_metrics, f, indent=4)

    def save_checkpoint(self, checkpoint_path):
        """trainer.py:319-335: {'model_state_dict', 'optimizer_state_dict', 'step'}"""
        self.flush()
        checkpoint = {
            'model_state_dict': self.nppc_model.state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
            'step': self.step,
        }
        os.makedirs(os.path.dirname(checkpoint_path) or ".", exist_ok=True)
        torch.save(checkpoint, checkpoint_path)
        print(f"Checkpoint saved to {checkpoint_path}")


class ShardedBatchSampler(torch.utils.data.Sampler):
    """Yields, for every global minibatch, this rank's contiguous slice of it (weak-scaling DP sharding)."""

    def __init__(self, n, global_batch, world, rank, shuffle, seed=0):
        self.n, self.gb, self.world, self.rank, self.shuffle, self.seed = n, global_batch, world, rank, shuffle, seed
        self.epoch = 0

    def __len__(self):
        # world > 1: the ragged tail is dropped so that every rank has equal work; one rank keeps it like the reference's
        # DataLoader(drop_last=False) does (nppc_audio/trainer.py:54-60): n_epochs -> steps then agree with the reference
        return self.n // self.gb if self.world > 1 else (self.n + self.gb - 1) // self.gb

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        self.epoch += 1
        per = self.gb // self.world
        if self.world == 1:
            for b0 in range(0, self.n, self.gb):
                yield order[b0:b0 + self.gb]
            return
        for b0 in range(0, self.n - self.gb + 1, self.gb):     # drop the ragged tail so every rank has equal work
            glob = order[b0:b0 + self.gb]
            yield glob[self.rank * per:(self.rank + 1) * per]
