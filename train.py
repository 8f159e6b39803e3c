"""Drop-in for the reference's ``train.py``: ``python train.py <experiment.yaml>`` (or ``accelerate launch``).

``ADTTrainer.compute_loss`` keeps the reference hook signature (train.py:40-78) so HF ``Trainer`` drives the
hand-written gfx950 forward/backward through autograd; ``--native`` instead runs the flat-buffer loop of
``adt_str_amd.trainer.FlatTrainer`` (one fused clip + AdamW launch, bucketed RCCL all-reduce overlapped with
backward, prefetched host pipeline, checkpoints + resume), which is what ``bench.py`` measures.

Multi-GPU, one process per GPU on one node, exactly like the reference's ``accelerate launch train.py <yaml>``
(README.md:53-57):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py <yaml> --native
    accelerate launch train.py <yaml> --native

    accelerate launch train.py <yaml>                  # HF Trainer + DistributedDataParallel, the reference's own command

All set RANK / LOCAL_RANK / WORLD_SIZE; ``train()`` binds the process to its GPU and creates the RCCL process group
before any GPU work (``adt_str_amd.trainer.init_distributed``), with or without ``--native``: HF Trainer's accelerate
state adopts that group and wraps the model in DDP over it.
"""
import argparse
import logging
import os
import random
from typing import Optional

import torch

from adt_str_amd.config_utils import load_merged
from adt_str_amd.masks import create_mask_plain
from adt_str_amd.network import ADTModel
from adt_str_amd.synth import SynthDrum, SynthDrumConfig
from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig
from build_model import model_config_from

try:
    from transformers import Trainer, TrainingArguments
except Exception:                                     # pragma: no cover - transformers is part of the image
    Trainer = object
    TrainingArguments = None


class ADTTrainer(Trainer):
    """HF Trainer subclass of the reference (train.py:33-78)."""

    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        # multi-GPU: `model` is the DDP wrapper accelerate built.  Tested on EVERY pass (an isinstance is free): an evaluation pass that
        # hands over the unwrapped model first (eval_on_start, evaluate() before train()) must not pin the slow path for the whole run.
        from torch.nn.parallel import DistributedDataParallel
        forward = model
        if isinstance(model, DistributedDataParallel):                       # (every pass: idempotent, and it follows a replaced engine object)
            from adt_str_amd.trainer import forward_engine_reduced, install_engine_reduction
            install_engine_reduction(model, getattr(self.args, "gradient_accumulation_steps", 1))
            forward = lambda **kw: forward_engine_reduced(model, **kw)        # (the engine reduces: DDP's bucket copies are skipped)
        model.train()
        device = next(model.parameters()).device
        tokens = inputs["tokens"].to(device)
        wavs = inputs["wavs"].to(device)
        token_lengths = inputs["token_lengths"].to(device)
        tgt_input, labels = tokens[:, :-1], tokens[:, 1:]                     # teacher forcing (train.py:56-57)
        _, tgt_padding_mask = create_mask_plain(tgt_input.size(1), token_lengths, device)
        loss = forward(src=wavs, tgt=tgt_input, tgt_mask=None, tgt_padding_mask=tgt_padding_mask, labels=labels)
        # the reference also runs gc.collect() + empty_cache() here every step (train.py:73-76): a host stall
        # and an allocator flush per step with no effect on the result -- deliberately not reproduced
        return (loss, None) if return_outputs else loss

    def create_optimizer(self):
        """The Trainer's AdamW (two groups: decay / no decay, ``get_decay_parameter_names``) on the engine's fused flat-buffer kernel
        when the run is the default ``adamw_torch*`` on a GPU (adt_str_amd/optim.py: the same arithmetic in one launch instead of
        multi-tensor passes over 132 tensors); ``ADT_HF_TORCH_ADAMW=1`` or any other ``--optim`` keeps transformers' own choice."""
        import os
        model = self.model
        use = (self.optimizer is None and os.environ.get("ADT_HF_TORCH_ADAMW") != "1" and str(getattr(self.args, "optim", "")).split(".")[-1].lower().startswith("adamw_torch")
               and hasattr(model, "engine") and next(model.parameters()).is_cuda and all(p.requires_grad for p in model.parameters()))
        if not use:
            return super().create_optimizer()
        from adt_str_amd.optim import FusedAdamW
        decay = set(self.get_decay_parameter_names(model))
        named = list(model.named_parameters())
        groups = [{"params": [p for n, p in named if n in decay], "weight_decay": self.args.weight_decay},
                  {"params": [p for n, p in named if n not in decay], "weight_decay": 0.0}]
        self.optimizer = FusedAdamW(groups, lr=self.args.learning_rate, betas=(self.args.adam_beta1, self.args.adam_beta2),
                                    eps=self.args.adam_epsilon, engine=model.engine)
        return self.optimizer

    def evaluate(self, eval_dataset=None, ignore_keys=None, metric_key_prefix="eval"):
        """Validation loss over an iterable of collated batches (reference train.py:80-141): the mean of the per-batch
        losses, logged as ``{prefix}_loss``; ``{}`` without an eval dataset (the shipped configs pass none, train.py:313).
        The per-batch ``.item()`` / ``gc.collect()`` / ``empty_cache()`` of the reference are not reproduced: the batch
        losses stay on the device and are read once at the end."""
        eval_dataset = eval_dataset if eval_dataset is not None else self.eval_dataset
        if eval_dataset is None:
            return {}
        model = self.model
        was_training = model.training
        model.eval()
        device = next(model.parameters()).device
        losses = []
        with torch.no_grad():
            for batch in eval_dataset:
                tokens = batch["tokens"].to(device)
                wavs = batch["wavs"].to(device)
                token_lengths = batch["token_lengths"].to(device)
                tgt_input, labels = tokens[:, :-1], tokens[:, 1:]
                _, tgt_padding_mask = create_mask_plain(tgt_input.size(1), token_lengths, device)
                losses.append(model(src=wavs, tgt=tgt_input, tgt_mask=None, tgt_padding_mask=tgt_padding_mask, labels=labels).reshape(()))
        model.train(was_training)
        avg_loss = float(torch.stack(losses).mean().item()) if losses else 0.0
        metrics = {f"{metric_key_prefix}_loss": avg_loss}
        self.log(metrics)
        return metrics


def create_training_arguments(cfg: dict) -> "TrainingArguments":
    from adt_str_amd.trainer import output_path
    t, lg, ex, ck = cfg["training"], cfg["logging"], cfg["experiment"], cfg["checkpoint"]
    kw = dict(output_dir=output_path(cfg),          # output_dir / run_name (reference train.py:171-176)
              per_device_train_batch_size=t["batch_size"], num_train_epochs=t["num_epochs"] or 1,
              learning_rate=t["learning_rate"], warmup_ratio=t["warmup_ratio"], weight_decay=t["weight_decay"],
              max_grad_norm=t["max_grad_norm"], gradient_accumulation_steps=t["gradient_accumulation_steps"], optim=t["optim"],
              lr_scheduler_type=t["lr_scheduler_type"], logging_steps=lg["logging_steps"], seed=ex["seed"],
              bf16=False,                     # the engine already computes in bf16 with fp32 accumulation; autocast has nothing to wrap
              dataloader_num_workers=0,       # the batch is rendered on the GPU inside collate: no CPU workers to feed
              dataloader_pin_memory=False,    # ... and it is already device memory (pinning a CUDA tensor raises)
              remove_unused_columns=False, report_to=[], save_total_limit=ck["max_checkpoints"],
              ddp_broadcast_buffers=False,     # PE tables / window / filterbank are constants (the reference re-broadcasts them each forward)
              ddp_find_unused_parameters=False,  # every parameter receives a gradient from _ADTLossFn.backward: no graph traversal per step
              save_strategy="steps" if lg.get("save_every_n_steps") else "epoch")
    if lg.get("save_every_n_steps"):
        kw["save_steps"] = lg["save_every_n_steps"]
    import inspect
    if "warmup_ratio" not in inspect.signature(TrainingArguments.__init__).parameters:
        kw["warmup_steps"] = kw.pop("warmup_ratio")       # transformers >= 5: a float < 1 in warmup_steps is the ratio
    if (t.get("lr_scheduler_type") or "cosine") == "cosine" and float(t.get("min_learning_rate") or 0) > 0:   # reference train.py:203-218
        kw["lr_scheduler_type"] = "cosine_warmup_with_min_lr"
        kw["lr_scheduler_kwargs"] = {"min_lr": float(t["min_learning_rate"])}
    return TrainingArguments(**kw)


def build_components(cfg: dict, device: str = "cuda"):
    shared = cfg["shared"]
    tokenizer = MidiTokenizer(MidiTokenizerConfig(**cfg["tokenizer"]))
    synth_cfg = dict(cfg["synthetiser"])
    synth_cfg.setdefault("ADTOF_mapping", cfg["tokenizer"]["ADTOF_mapping"])
    synth = SynthDrum(SynthDrumConfig(**shared, **synth_cfg), device=device)
    model = ADTModel(model_config_from(cfg))
    return model, tokenizer, synth


def train(cfg: dict, native: bool = False):
    from adt_str_amd.trainer import init_distributed, run_native_training
    from data_modules.train_dataset import LakhDataset, LakhDatasetConfig
    logging.basicConfig(level=getattr(logging, cfg["logging"].get("log_level", "INFO")))
    # One process per GPU under ``accelerate launch`` / ``torchrun``: bind this process to cuda:LOCAL_RANK and create the RCCL
    # process group BEFORE any GPU work -- for the native loop and for HF Trainer alike (accelerate's PartialState adopts an
    # already-initialised process group instead of creating its own, and DDP then runs on it).  A no-op for a single process.
    rank, local_rank, world = init_distributed()
    device = f"cuda:{local_rank}" if world > 1 else "cuda"
    seed = cfg["experiment"]["seed"]
    random.seed(seed)
    torch.manual_seed(seed)
    model, tokenizer, synth = build_components(cfg, device=device)
    ds = LakhDataset(LakhDatasetConfig(**cfg["shared"], **cfg["TrainDatasetConfig"]), tokenizer, synth)
    dump = os.environ.get("ADT_DUMP_PARAMS")                  # test hook: every rank's final parameters
    if native:
        tr = run_native_training(model.to(device), ds, cfg)
        if dump:
            torch.cuda.synchronize()
            torch.save({"pflat": tr.pflat.detach().cpu(), "steps": tr.step_no, "world": tr.world}, os.path.join(dump, f"params_rank{tr.rank}.pt"))
        return tr
    # HF Trainer + DDP (the reference's own launch, README.md:53-57): ranks draw different dropout masks, as nn.Dropout does under DDP
    model.seed_dropout(seed, rank)
    trainer = ADTTrainer(model=model, args=create_training_arguments(cfg), train_dataset=ds, data_collator=ds.collate)
    trainer.train(resume_from_checkpoint=cfg["checkpoint"].get("resume_from_checkpoint"))
    trainer.save_model()
    if dump:
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1).float().cpu() for _, p in model.named_parameters()])
        torch.save({"pflat": flat, "steps": trainer.state.global_step, "world": world}, os.path.join(dump, f"params_rank{rank}.pt"))
    return trainer


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=str)
    ap.add_argument("--native", action="store_true", help="flat-buffer training loop instead of HF Trainer")
    a = ap.parse_args()
    train(load_merged(a.config), native=a.native)
