"""`Trainer` — the reference's incremental-learning trainer (`Trainer.py`, class at :100) with the hot loops on the
cxrk HIP kernels.  Same constructor, method names, argument meaning, module-level switches and error behaviour, so
`ZERO_JOINT_BOUNDS.py` / `CLASS_INCREMENTAL.py` / `DATA_INCREMENTAL.py`-style drivers run unchanged.

What changed underneath (results identical):
  * the 10 frozen CXR-BERT calls per step embed constant strings under no_grad (`Trainer.py:557-567,1657-1680`,
    `text/inference_engine.py:50`): their outputs are computed once and cached;
  * the 5x(pos,neg) Python loop of tiny cosine launches becomes one [B,128]x[128,2C] cosine kernel, and
    `nn.BCEWithLogitsLoss()` on (pos - neg) becomes one fused loss+gradient kernel (any other criterion is applied
    to the logits tensor as given);
  * `optim.Adam` / `optim.SGD` become one fused flat-buffer kernel; `myIncremental` a two-kernel reduction+restore.
Plotting / t-SNE / heat-map reporting (`Trainer.py:1074-1185,1310-1554`) is host-side visualisation and is not part
of this module; scalar metrics (sklearn) are kept.

Two extensions behind the same entry points (both off unless asked for):
  * `Trainer(..., joint_encoders={"image_model": ImageModel, "temperature": 0.07})` — the north-star step: both encoders train
    in-loop.  Loaders then yield `(images [B,3,H,W], input_ids [B,L], attention_mask [B,L][, labels [B,5]])`; `train`,
    `train_class_incremental` and `train_class_more_labels_incremental` run `contrastive.JointContrastiveTrainer.step` (ResNet-50 +
    CXR-BERT forward, InfoNCE over the global batch, hand-written backward, fused Adam) per batch, with the reference's
    iteration / logging / continual-learning bookkeeping around it; `val` / `test` embed the images with the trained image
    model and score them against the class prompts embedded by the trained text model (the zero-shot chain of
    `Trainer.py:797-837`).  The text model is the one inside `bert_encoder`.
  * data parallelism: with `torch.distributed` initialised (one process per GPU) every rank draws the SAME global batch from its
    loader (same sampler seed) and trains on its contiguous row shard; adapter gradients (0.5 MB) are summed with one
    all-reduce of the flat gradient buffer, weighted by shard size, so the update equals the single-process global-batch one
    (SURVEY.md §8e).  In joint mode the shards' embeddings are all-gathered for the global similarity matrix.
"""
from __future__ import annotations

import copy
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import ConcatDataset, DataLoader, RandomSampler, Subset, TensorDataset

from . import functional as Fh
from . import kernels as K
from . import optim as cxr_optim
from .DataRetrieval import CHEXPERT_COMPETITION_CLASSES, basic_create_prompts, create_prompts
from .health_multimodal.text import get_cxr_bert_inference
from .models import myLinearModel, myMLP

# module-level switches, same names / defaults as the reference (Trainer.py:39-56)
SHARED = False
IMAGE_MODEL = True
TEXT_MODEL = True
MODEL_USED = "mlp"  # mlp, dense, "no-head"
OPTIM = "adam"  # sgd
MAX_EMB = False
NEW_PROMPTS = False
TRAIN_LOGIT_DIFF = True   # False <--> only POS
PRED_LOGIT_DIFF = False   # False <--> only POS
CHANGE_LABELS = False


class ScalarWriter:
    """Minimal stand-in for `torch.utils.tensorboard.SummaryWriter` (tensorboard is not installed here): keeps
    `add_scalar(tag, value, step)` and `log_dir`; values are buffered on the device and written as JSON lines on
    `flush()/close()` so logging never forces a host sync inside the training loop."""

    def __init__(self, log_dir: str):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self._rows: List[tuple] = []

    def add_scalar(self, tag, value, step=None):
        self._rows.append((tag, value.detach() if isinstance(value, torch.Tensor) else float(value), step))

    def add_figure(self, *a, **k):
        pass

    def scalars(self, tag=None):
        return [(t, float(v), s) for t, v, s in self._rows if tag is None or t == tag]

    def flush(self):
        import json
        with open(os.path.join(self.log_dir, "scalars.jsonl"), "a") as f:
            for t, v, s in self._rows:
                f.write(json.dumps({"tag": t, "value": float(v), "step": s}) + "\n")
        self._rows = []

    close = flush


def _make_writer(path: str):
    try:
        from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433
        return SummaryWriter(path)
    except Exception:
        return ScalarWriter(path)


@torch.no_grad()
def change_values(tensor):
    """`Trainer.py:1708-1728`: as shipped maps 1 -> 1 and 0 -> -1, float32."""
    new_tensor = tensor.clone()
    new_tensor[tensor == 1] = 1
    new_tensor[tensor == 0] = -1
    return new_tensor.float().to(tensor.device)


class Trainer:
    def __init__(self, single_prompt, prompts, class_names, loss_name, lr, device, writer, bert_encoder=None, joint_encoders=None,
                 process_group=None):
        """Same positional arguments as the reference (`Trainer.py:101`).  `bert_encoder` (optional) injects a
        `TextInferenceEngine`; default `get_cxr_bert_inference()` as at `Trainer.py:109` (Hub fetch, or the offline
        synthetic model when CXRK_SYNTHETIC_WEIGHTS=1).  `joint_encoders` (optional, see the module docstring): a dict with
        "image_model" (an `ImageModel` on `device`) and optionally "temperature" (default 0.07) — no adapters are created, the
        trainable parameters are the two encoders.  `process_group`: the data-parallel group (default: the default group when
        `torch.distributed` is initialised)."""
        self.pos_mean_counter = 0
        self.neg_mean_counter = 0
        self.n_reset = 0
        self.n_updated = 0
        self.image_adapter_copy = None
        self.text_adapter_copy = None
        self.bert_encoder = bert_encoder if bert_encoder is not None else get_cxr_bert_inference()
        self.bert_encoder.to(device)
        self.prompts = prompts
        self.class_names = class_names
        self.device = device
        self.writer = writer
        self.loss_name = loss_name
        self.change_labels = CHANGE_LABELS
        if self.change_labels:
            print("*** Watch out! Changing labels is enabled! ***")
        self.basic_prompts = single_prompt
        print("Single prompt per class" if single_prompt else "Multiple prompts per class")
        print("*** LOSS " + str(loss_name) + " ***")

        def new_adapter():
            if MODEL_USED == "mlp":
                return myMLP().to(device)
            if MODEL_USED == "dense":
                return myLinearModel().to(device)
            print("*** ERROR... ***")
            raise Exception

        import torch.distributed as dist
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self._joint = None
        self.image_model = None
        if joint_encoders is not None:
            from .contrastive import JointContrastiveTrainer
            je = dict(joint_encoders) if isinstance(joint_encoders, dict) else {"image_model": joint_encoders[0], "temperature": joint_encoders[1]}
            self.image_model = je["image_model"]
            if OPTIM not in ("adam", "sgd"):
                raise Exception
            self._joint = JointContrastiveTrainer(self.image_model, self.bert_encoder.model, lr=lr,
                                                  temperature=float(je.get("temperature", 0.07)), group=process_group, optim=OPTIM)
            print("*** JOINT ENCODER TRAINING (InfoNCE over the global batch): no adapters ***")
        params = []
        if self._joint is not None:
            self.image_adapter = self.text_adapter = None
        elif SHARED:
            print("*** SHARED MODEL !!!! ***")
            shared_model = new_adapter()
            self.image_adapter = shared_model
            self.text_adapter = shared_model
            params += list(shared_model.parameters())
        else:
            self.text_adapter = None
            if TEXT_MODEL:
                self.text_adapter = new_adapter()
                params += self.text_adapter.parameters()
            else:
                print("*** No text adapter !!!! ***")
            self.image_adapter = None
            if IMAGE_MODEL:
                self.image_adapter = new_adapter()
                params += self.image_adapter.parameters()
            else:
                print("*** No IMAGE MODEL !!!! ***")
        self._has_img = self.image_adapter is not None
        self._has_txt = self.text_adapter is not None
        print("image adapter", self.image_adapter)
        print("text adapter", self.text_adapter)
        if self._joint is not None:
            self.optimizer = self._joint.optimizer
        elif len(params) > 0:
            if OPTIM == "adam":
                print("Creating Adam optimizer...")
                self.optimizer = cxr_optim.Adam(params, lr=lr)
            elif OPTIM == "sgd":
                print("Creating SGD optimizer...")
                self.optimizer = cxr_optim.SGD(params, lr=lr)
            else:
                raise Exception
        else:
            self.optimizer = None
        if self.world > 1 and self.optimizer is not None:
            # replicas must start identical: rank 0's initial weights everywhere (the drivers seed every rank alike, this makes it
            # independent of that)
            dist.broadcast(self.optimizer.flat_p, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
        self.val_f1_heat_map = torch.empty((0, 5))
        self.val_auroc_heat_map = torch.empty((0, 5))
        self.test_f1_heat_map = torch.empty((0, 5))
        self.test_auroc_heat_map = torch.empty((0, 5))
        self._bert_cache: Dict[tuple, torch.Tensor] = {}
        self._flat_copy = None
        self._reset_counters = torch.zeros(2, dtype=torch.int64, device=device)
        self._reset_total = 0

    # ------------------------------------------------------------------------------------------------ data
    @staticmethod
    def _preprocessing(chex_competition, xrays_position, batch_size, dataset_root: Optional[str] = None):
        """Loaders over pre-computed `[N,128]` image embeddings + `[N,5]` labels (`Trainer.py:199-253`).
        The reference reads `embeddingDataset\\{train,val,test}\\512-chex-not-normalize[-frontal]\\
        embeddings_dataset_final_old.pt` relative to the CWD; the same layout is looked up under `dataset_root`
        (or $CXRK_EMBEDDING_DATASET).  Those files are whole-object pickles: they are opened with
        `weights_only=True` and must contain tensors only (a dict {"embs","labels"} or a tuple)."""
        if not chex_competition:
            raise Exception
        print("*** CHEX COMPETITION ***")
        class_names = list(CHEXPERT_COMPETITION_CLASSES)
        chex_str = "-chex"
        if xrays_position == "all":
            sub = "512" + chex_str + "-not-normalize"
        elif xrays_position == "frontal":
            sub = "512" + chex_str + "-not-normalize-frontal"
        else:
            raise Exception
        root = dataset_root or os.environ.get("CXRK_EMBEDDING_DATASET", "embeddingDataset")

        def load(split):
            path = os.path.join(root, split, sub, "embeddings_dataset_final_old.pt")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path}: pre-computed embedding dataset not found (Trainer.py:221-235); "
                                        f"use Trainer.synthetic_loaders(...) for synthetic data")
            obj = torch.load(path, map_location="cpu", weights_only=True)
            embs, labels = (obj["embs"], obj["labels"]) if isinstance(obj, dict) else obj
            return TensorDataset(embs.float(), labels.float())

        train_dataset, val_dataset, test_dataset = load("train"), load("val"), load("test")
        print("TrainBS:", batch_size, "Val/Test Batch size default set to 1024")
        mk = lambda ds, bs: DataLoader(ds, sampler=None, batch_size=bs, shuffle=True, num_workers=0, pin_memory=True,  # noqa: E731
                                       drop_last=False)
        return class_names, chex_str, mk(train_dataset, batch_size), mk(val_dataset, 1024), mk(test_dataset, 1024), None

    @staticmethod
    def synthetic_loaders(n_train: int, n_val: int, n_test: int, batch_size: int, seed: int = 29, shuffle: bool = True):
        """Synthetic stand-in for `_preprocessing`: N(0,1) embeddings, Bernoulli(0.3) labels."""
        from .synthetic import synthetic_adapter_batch
        out = []
        for i, (n, bs) in enumerate(((n_train, batch_size), (n_val, 1024), (n_test, 1024))):
            e, l, _ = synthetic_adapter_batch(n, seed=seed + i)
            g = torch.Generator().manual_seed(seed + 100 + i)
            out.append(DataLoader(TensorDataset(e, l), batch_size=bs, shuffle=shuffle, generator=g, num_workers=0,
                                  pin_memory=False, drop_last=False))
        return out

    @staticmethod
    def synthetic_joint_loaders(n_train: int, n_val: int, n_test: int, batch_size: int, image_size: int = 224, seq_len: int = 32,
                                vocab: int = 30522, seed: int = 27, eval_batch_size: int = 1024):
        """Loaders for the joint-encoder mode (SURVEY.md §8d inputs): `(images [B,3,S,S] in [0,1) with one plane replicated to 3
        channels, token ids [B,L], attention mask [B,L], multi-hot labels [B,5])`."""
        from .synthetic import synthetic_images, synthetic_tokens
        out = []
        for i, (n, bs) in enumerate(((n_train, batch_size), (n_val, eval_batch_size), (n_test, eval_batch_size))):
            ids, mask = synthetic_tokens(n, seq_len, vocab=vocab, seed=seed + 1 + 10 * i)
            g = torch.Generator().manual_seed(seed + 2 + 10 * i)
            labels = (torch.rand(n, 5, generator=g) < 0.3).float()
            ds = TensorDataset(synthetic_images(n, image_size, seed=seed + 10 * i), ids, mask, labels)
            gs = torch.Generator().manual_seed(seed + 100 + i)
            out.append(DataLoader(ds, batch_size=bs, shuffle=True, generator=gs, num_workers=0, pin_memory=False, drop_last=False))
        return out

    @staticmethod
    def _run_name(prefix, loss_name, lr, batch_size, epochs, chex_str, str_basic, xrays_position, extra=""):
        suffix = "-" + MODEL_USED
        if SHARED:
            suffix += "-SHARED-adapter"
        elif IMAGE_MODEL and TEXT_MODEL:
            suffix += "-double-adapter"
        elif IMAGE_MODEL:
            suffix += "-only-image-adapter"
        elif TEXT_MODEL:
            suffix += "-only-text-adapter"
        return (prefix + "-loss-" + str(loss_name) + "-opt-" + OPTIM + "-lr-" + str(lr) + "-bs" + str(batch_size) + "-ep"
                + str(epochs) + extra + chex_str + str_basic + "-" + str(xrays_position) + suffix)

    @staticmethod
    def _prompts_for(class_names, single_prompt):
        if single_prompt:
            return "-single-prompt", basic_create_prompts(class_names)
        return ("-MAX-prompt" if MAX_EMB else "-mean-prompt"), create_prompts(class_names, NEW_PROMPTS, TRAIN_LOGIT_DIFF)

    @staticmethod
    def _logit_suffix(w_path):
        w_path += "-NEW-PROMPTS" if NEW_PROMPTS else ""
        w_path += "-TRAIN-logit-DIFF" if TRAIN_LOGIT_DIFF else "-TRAIN-logit-POS"
        w_path += "-PRED-logit-DIFF" if PRED_LOGIT_DIFF else "-PRED-logit-POS"
        return w_path

    @staticmethod
    def preprocessing(chex_competition, xrays_position, single_prompt, batch_size, lr, epochs, loss_name,
                      dataset_root=None, log_root="."):
        """`Trainer.py:256-328`: returns (writer, class_names, train_loader, val_loader, test_loader, prompts,
        plot_tsne_array)."""
        class_names, chex_str, train_loader, val_loader, test_loader, plot_tsne_array = Trainer._preprocessing(
            chex_competition, xrays_position, batch_size, dataset_root)
        folder_name = "NUOVI_RISULTATI-3/zero-and-joint"
        str_basic, prompts = Trainer._prompts_for(class_names, single_prompt)
        if epochs > 0:
            w_path = os.path.join(log_root, folder_name, Trainer._run_name("joint-train", loss_name, lr, batch_size, epochs,
                                                                           chex_str, str_basic, xrays_position))
        else:
            if SHARED and IMAGE_MODEL and TEXT_MODEL:
                suffix = "-SHARED-adapter-" + MODEL_USED
            elif not SHARED and not IMAGE_MODEL and not TEXT_MODEL:
                suffix = "-no-head"
            else:
                raise Exception
            print("Attenzione! Zero-shot evaluation!")
            w_path = os.path.join(log_root, folder_name, "zero-shot-model" + chex_str + str_basic + "-" + str(xrays_position) + suffix)
        w_path = Trainer._logit_suffix(w_path)
        print("writer path:", w_path)
        return _make_writer(w_path), class_names, train_loader, val_loader, test_loader, prompts, plot_tsne_array

    @staticmethod
    def preprocessing_class_incremental(chex_competition, xrays_position, basic_prompts, batch_size, lr, epochs, loss_name,
                                        mode, CONTINUAL_LEARNING=None, ratio=None, threshold=None,
                                        threshold_scheduling=False, adder=0.01, MORE_LABELS=False, dataset_root=None,
                                        log_root="."):
        """`Trainer.py:330-435`: train_loader becomes a list of 5 task loaders."""
        print("**** Gradient Clipping ****\n--->" + str(CONTINUAL_LEARNING) if CONTINUAL_LEARNING is not None
              else "**** NO Gradient Clipping ****")
        class_names, chex_str, train_loader, val_loader, test_loader, plot_tsne_array = Trainer._preprocessing(
            chex_competition, xrays_position, batch_size, dataset_root)
        folder_name = "NUOVI_RISULTATI/" + mode + ("-more-labels" if MORE_LABELS else "")
        if mode == "class-pos-neg":
            train_loader = Trainer.split_dataloader_data_incremental(Trainer.concat_to_tensor_dataloader(train_loader), 5)
        elif mode == "class-pos":
            train_loader = Trainer.split_dataloader_by_label(Trainer.concat_to_tensor_dataloader(train_loader), batch_size=batch_size)
        else:
            raise Exception
        if epochs == 0:
            raise Exception
        thre_str = "-th-scheduled-" + str(adder) if (threshold_scheduling and CONTINUAL_LEARNING is not None) else ""
        cl_str = ""
        if CONTINUAL_LEARNING is not None and ratio:
            cl_str = "-" + str(CONTINUAL_LEARNING) + "-ratio-" + str(threshold)
            mode = "gradient-clipping-" + mode
        else:
            mode = "fine-tuning-" + mode
        str_basic, prompts = Trainer._prompts_for(class_names, basic_prompts)
        w_path = os.path.join(log_root, folder_name, Trainer._run_name(mode, loss_name, lr, batch_size, epochs, chex_str,
                                                                       str_basic, xrays_position)) + cl_str + thre_str
        w_path += "-MORE-LABELS" if MORE_LABELS else ""
        w_path = Trainer._logit_suffix(w_path) + "-DD"
        print("writer path:", w_path)
        return _make_writer(w_path), class_names, train_loader, val_loader, test_loader, prompts, plot_tsne_array

    @staticmethod
    def preprocessing_data_incremental(chex_competition, xrays_position, basic_prompts, batch_size, lr, parts, epochs,
                                       loss_name, mode, CONTINUAL_LEARNING=None, ratio=None, threshold=None,
                                       threshold_scheduling=False, adder=0.01, dataset_root=None, log_root="."):
        """`Trainer.py:437-523`: train_loader becomes a list of `parts` contiguous shards."""
        folder_name = "NUOVI_RISULTATI/data-incremental-" + str(parts) + "-parts"
        class_names, chex_str, train_loader, val_loader, test_loader, plot_tsne_array = Trainer._preprocessing(
            chex_competition, xrays_position, batch_size, dataset_root)
        if mode != "data-inc":
            raise Exception
        print("number of parts:", parts)
        train_loader = Trainer.split_dataloader_data_incremental(train_loader, parts)
        if epochs == 0:
            raise Exception
        thre_str = "-th-scheduled-" + str(adder) if (CONTINUAL_LEARNING is not None and threshold_scheduling) else ""
        cl_str = ""
        if CONTINUAL_LEARNING is not None and ratio:
            cl_str = "-" + str(CONTINUAL_LEARNING) + "-ratio-" + str(threshold)
            mode = "gradient-clipping-" + mode
        else:
            mode = "fine-tuning-" + mode
        str_basic, prompts = Trainer._prompts_for(class_names, basic_prompts)
        w_path = os.path.join(log_root, folder_name, Trainer._run_name(mode, loss_name, lr, batch_size, epochs, chex_str,
                                                                       str_basic, xrays_position,
                                                                       extra="-parts" + str(parts))) + cl_str + thre_str
        w_path = Trainer._logit_suffix(w_path) + "-DD"
        print("writer path:", w_path)
        return _make_writer(w_path), class_names, train_loader, val_loader, test_loader, prompts, plot_tsne_array

    # ------------------------------------------------------------------------------------------------ splitters
    @staticmethod
    def split_dataloader_by_label(dataloader, batch_size):
        """One loader per label with the samples positive for it (`Trainer.py:1187-1212`)."""
        if not isinstance(dataloader.dataset, TensorDataset):
            raise ValueError("Unsupported dataset type")
        loaders = []
        for i in range(5):
            indices = torch.where(dataloader.dataset.tensors[-1][:, i] == 1)[0]   # labels: the last tensor ([1] in the reference's pairs)
            subset = Subset(dataloader.dataset, indices)
            loaders.append(DataLoader(subset, batch_size=batch_size, sampler=RandomSampler(subset), num_workers=0,
                                      pin_memory=True, drop_last=False))
        return loaders

    @staticmethod
    def split_dataloader_data_incremental(dataloader, n):
        """N contiguous, equally sized shards (`Trainer.py:1214-1231`)."""
        dataset = dataloader.dataset
        num_samples = len(dataset)
        subset_size = math.ceil(num_samples / n)
        subsets = [Subset(dataset, range(i * subset_size, min((i + 1) * subset_size, num_samples))) for i in range(n)]
        return [DataLoader(s, batch_size=dataloader.batch_size, sampler=RandomSampler(s), num_workers=0, pin_memory=True,
                           drop_last=False) for s in subsets]

    @staticmethod
    def concat_to_tensor_dataloader(dataloader):
        """ConcatDataset of TensorDatasets -> one TensorDataset loader (`Trainer.py:1253-1271`)."""
        ds = dataloader.dataset
        if isinstance(ds, TensorDataset):
            return dataloader
        cols = [torch.cat([d.tensors[j] for d in ds.datasets]) for j in range(len(ds.datasets[0].tensors))]
        return DataLoader(TensorDataset(*cols), batch_size=dataloader.batch_size,
                          num_workers=0, pin_memory=dataloader.pin_memory, drop_last=dataloader.drop_last)

    @staticmethod
    def count_positive_labels(dataloader):
        tot = torch.zeros(5)
        for batch in dataloader:
            tot += batch[-1].sum(0)
        for i in range(5):
            print(f"Label {i}: {tot[i]}")

    # ------------------------------------------------------------------------------------------------ text side
    @torch.no_grad()
    def _bert_embed(self, prompts: Sequence[str]) -> torch.Tensor:
        """Frozen CXR-BERT embeddings of a prompt list, un-normalised (`Trainer.py:1660`); cached, since the encoder
        is frozen, in eval mode and called under no_grad (`text/inference_engine.py:50,63`)."""
        key = tuple(prompts)
        hit = self._bert_cache.get(key)
        if hit is None:
            hit = self.bert_encoder.get_embeddings_from_prompt(list(prompts), normalize=False, verbose=False).to(self.device)
            self._bert_cache[key] = hit   # (joint mode: emptied by every training step, the text encoder is no longer a constant)
        return hit

    def bert_forward_mean(self, pos_prompt, neg_prompt, use_grad, to_plot=False):
        """`Trainer.py:1657-1680`: BERT embeddings -> text adapter -> mean over prompts (unless single prompt / MAX_EMB)."""
        out = []
        with torch.set_grad_enabled(use_grad):
            for prompt in (pos_prompt, neg_prompt):
                e = self._bert_embed(prompt)
                if self._has_txt:
                    e = self.text_adapter(e)
                assert e.shape[0] == len(prompt)
                if (not self.basic_prompts and not MAX_EMB) or to_plot:
                    e = Fh.group_mean(e, 1, e.shape[0]).reshape(-1)
                out.append(e)
        return out[0], out[1]

    def myCosineSimilarity(self, x, y, use_grad, to_plot=False, train=False, pos=None):
        """`Trainer.py:1682-1704`: torchmetrics pairwise cosine of x [B,128] against y ([128] -> [1,128], or the
        un-averaged prompt set when MAX_EMB: max over prompts)."""
        with torch.set_grad_enabled(use_grad):
            if to_plot:
                return Fh.pairwise_cosine_similarity(x.reshape(1, -1), y.reshape(1, -1))
            if not MAX_EMB:
                return Fh.pairwise_cosine_similarity(x, y.reshape(1, -1))
            res, res_mean, _ = Fh.pairwise_cosine_max(x, y.reshape(-1, x.shape[1]), 1)
            res, res_mean = res.reshape(-1), res_mean.reshape(-1)
            if train and self.writer is not None:
                gap = float((res.detach().double() - res_mean.double()).mean())   # one scalar for the log, off the gradient path
                if pos:
                    self.pos_mean_counter += 1
                    self.writer.add_scalar("max-mean-comparison/pos", gap, self.pos_mean_counter)
                else:
                    self.neg_mean_counter += 1
                    self.writer.add_scalar("max-mean-comparison/neg", gap, self.neg_mean_counter)
            return res

    def _prompt_matrix(self, class_names: Sequence[str], use_grad: bool) -> torch.Tensor:
        """[2C,128]: row 2c / 2c+1 = positive / negative prompt vector of class c after the text adapter and the
        mean over prompts — the whole per-class loop of `Trainer.py:557-567` in three launches."""
        groups, n = [], None
        for c in class_names:
            pos = self.prompts[c]["positive"]
            neg = self.prompts[c]["negative"] if TRAIN_LOGIT_DIFF else self.prompts[c]["positive"]
            for pr in (pos, neg):
                if n is None:
                    n = len(pr)
                if len(pr) != n:
                    return None  # ragged prompt sets: caller falls back to the per-class path
                groups.append(self._bert_embed(pr))
        with torch.set_grad_enabled(use_grad):
            e = torch.cat(groups, dim=0)
            if self._has_txt:
                e = self.text_adapter(e)
            if MAX_EMB:
                return e   # [2C*n,128]: the max over each group's n prompts is taken on the cosines (`Trainer.py:1691-1693`)
            return Fh.group_mean(e, len(groups), n) if n > 1 else e

    def _logits_and_loss(self, new_embs, labels, class_names, criterion, use_grad):
        """logits [B,C] (+ loss).  Fast path: fused cosine + BCE kernels; otherwise the reference's per-class loop."""
        pm = self._prompt_matrix(class_names, use_grad)
        with torch.set_grad_enabled(use_grad):
            if pm is not None:
                if MAX_EMB:
                    cos, cos_mean, _ = Fh.pairwise_cosine_max(new_embs, pm, 2 * len(class_names))
                    if use_grad and self.writer is not None:   # the reference's per-call log (`Trainer.py:1694-1703`)
                        gap = (cos.detach().double() - cos_mean.double()).mean(0).tolist()
                        for g, v in enumerate(gap):
                            tag, ctr = ("pos", "pos_mean_counter") if g % 2 == 0 else ("neg", "neg_mean_counter")
                            setattr(self, ctr, getattr(self, ctr) + 1)
                            self.writer.add_scalar("max-mean-comparison/" + tag, v, getattr(self, ctr))
                else:
                    cos = Fh.pairwise_cosine_similarity(new_embs, pm)
                # the fused loss + gradient kernel serves `nn.BCEWithLogitsLoss` with reduction "mean" (the reference's criterion) or
                # "sum"; per-element weights, pos_weight, reduction "none" or any other criterion module are applied to the logits
                # tensor as the caller's module defines them
                fused = (type(criterion) is nn.BCEWithLogitsLoss and criterion.weight is None
                         and criterion.pos_weight is None and criterion.reduction in ("mean", "sum") and not self.change_labels)
                if fused:
                    lab = labels if labels.dim() == 2 else labels.unsqueeze(1)
                    loss, logits = Fh.posneg_bce_loss(cos, lab, TRAIN_LOGIT_DIFF, criterion.reduction)
                    if labels.dim() == 1:
                        logits = logits.reshape(-1)
                    return logits, loss, cos
                logits = (cos[:, 0::2] - cos[:, 1::2]) if TRAIN_LOGIT_DIFF else cos[:, 0::2]
            else:
                cols, cosl = [], []
                for c in class_names:
                    pos = self.prompts[c]["positive"]
                    neg = self.prompts[c]["negative"] if TRAIN_LOGIT_DIFF else pos
                    pe, ne = self.bert_forward_mean(pos, neg, use_grad=use_grad)
                    ps = self.myCosineSimilarity(new_embs, pe, use_grad=use_grad, train=use_grad, pos=True)
                    ns = self.myCosineSimilarity(new_embs, ne, use_grad=use_grad, train=use_grad, pos=False)
                    cols.append(ps.flatten() - ns.flatten() if TRAIN_LOGIT_DIFF else ps.flatten())
                    cosl += [ps.flatten(), ns.flatten()]
                logits = torch.stack(cols, dim=1)
                cos = torch.stack(cosl, dim=1)
            if labels.dim() == 1:
                logits = logits.reshape(-1)
            lab = change_values(labels) if self.change_labels else labels
            loss = criterion(logits, lab) if criterion is not None else None
            return logits, loss, cos

    # ------------------------------------------------------------------------------------------------ hot loops
    def _set_mode(self, train: bool):
        """adapters follow the loop (`Trainer.py:533-535,779-782`); in joint mode the encoders stay in eval mode throughout —
        BatchNorm on running statistics, dropout off: the only mode the HIP path implements (parameters still get gradients)"""
        for m in (self.image_adapter, self.text_adapter):
            if m is not None:
                m.train(train)

    def _shard(self, n: int):
        """this rank's contiguous row range of a global batch of n rows (tensor_split boundaries)"""
        return (self.rank * n) // self.world, ((self.rank + 1) * n) // self.world

    def _train_step(self, batch, class_names, criterion, label_cols=None):
        """One optimisation step on one loader batch.  Reference form: batch = (embs, labels), `label_cols` picks the label
        column(s) of the task.  Joint form: batch = (images, input_ids, attention_mask[, labels])."""
        if self.loss_name != "standard":
            raise Exception
        if self._joint is not None:
            return self._joint_step(batch)
        embs, labels = batch
        if label_cols is not None:
            labels = labels[:, label_cols]
        n = embs.shape[0]
        lo, hi = self._shard(n) if self.world > 1 else (0, n)
        self.optimizer.zero_grad()
        if hi > lo:
            embs = embs[lo:hi].to(self.device, non_blocking=True)
            labels = labels[lo:hi].to(self.device, non_blocking=True)
            new_embs = self.image_adapter(embs) if self._has_img else embs
            logits, loss, _ = self._logits_and_loss(new_embs, labels, class_names, criterion, use_grad=True)
            loss.backward()
        else:   # a ragged last batch with fewer rows than ranks: this rank contributes nothing, but takes part in the collectives
            loss = torch.zeros((), dtype=torch.float32, device=self.device)
        if self.world > 1:
            # mean over the GLOBAL batch = sum over ranks of (rows of the shard / rows of the batch) x shard mean
            import torch.distributed as dist
            w = (hi - lo) / float(n)
            self.optimizer.all_reduce_grads(self.group, pre_scale=w)
            loss = K.scale_mask(loss.detach().reshape(1), alpha=w).reshape(())
            dist.all_reduce(loss, group=self.group)
        self.optimizer.step()
        return loss.detach()

    def _joint_step(self, batch):
        """The north-star step on one batch `(images, input_ids, attention_mask[, labels])`: this rank's row shard through both
        encoders, InfoNCE over the global batch, backward, gradient all-reduce, fused optimiser step."""
        if len(batch) < 3:
            raise ValueError("joint-encoder training expects loaders that yield (images, input_ids, attention_mask[, labels]); got a "
                             f"batch of {len(batch)} tensors")
        images, ids, mask = batch[0], batch[1], batch[2]
        n = images.shape[0]
        lo, hi = self._shard(n) if self.world > 1 else (0, n)
        if self.world > 1 and n % self.world:
            raise ValueError(f"joint-encoder training shards the batch evenly: {n} rows over {self.world} ranks (use drop_last)")
        dev = self.device
        loss = self._joint.step(images[lo:hi].to(dev, non_blocking=True), ids[lo:hi].to(dev, non_blocking=True),
                                mask[lo:hi].to(dev, non_blocking=True))
        self._bert_cache.clear()
        return loss

    def train(self, train_loader, criterion, epoch, CONTINUAL_LEARNING=None, threshold=None, scheduler=None, part=None,
              epochs=None, actual_task=None):
        """One epoch of joint / data-incremental training (`Trainer.py:526-606`)."""
        batch_idx = 0
        iteration = 0
        self._set_mode(True)
        cl = CONTINUAL_LEARNING == "myCL" and actual_task is not None and actual_task > 1
        for batch in train_loader:
            if cl:
                self.model_copy()
            batch_idx += 1
            loss = self._train_step(batch, self.class_names, criterion)
            if part is None:
                iteration = (epoch - 1) * len(train_loader) + batch_idx
            else:
                iteration = (part - 1) * epochs * len(train_loader) + (epoch - 1) * len(train_loader) + batch_idx
            if cl:
                self.myIncremental(threshold, iteration)
            if self.writer is not None:
                self.writer.add_scalar('train/Loss', loss, iteration)
            if scheduler is not None:
                scheduler.step()
                self.writer.add_scalar('train/LR', self.optimizer.param_groups[0]['lr'], iteration)
        if cl:
            self.myIncremental_save_log(iteration)

    def train_class_incremental(self, train_loader, criterion, epoch, CONTINUAL_LEARNING=None, threshold=None,
                                current_task=None, last_batch=0, actual_task=None):
        """One epoch on the single label column `current_task` (`Trainer.py:608-680`); returns the running iteration.  (Joint-encoder
        mode: the InfoNCE step has no label columns — the task is defined by what its loader yields, `current_task` only numbers it.)"""
        batch_idx = last_batch
        self._set_mode(True)
        cl = CONTINUAL_LEARNING == "myCL" and actual_task is not None and actual_task > 1
        names = [self.class_names[current_task]]
        for batch in train_loader:
            if cl:
                self.model_copy()
            batch_idx += 1
            loss = self._train_step(batch, names, criterion, label_cols=current_task)
            if cl:
                self.myIncremental(threshold, batch_idx)
            if self.writer is not None:
                self.writer.add_scalar('train/Loss', loss, batch_idx)
        if cl:
            self.myIncremental_save_log(batch_idx)
        return batch_idx

    def train_class_more_labels_incremental(self, train_loader, criterion, epoch, CONTINUAL_LEARNING=None, threshold=None,
                                            current_task=None, last_batch=0, actual_task=None):
        """One epoch on label columns `[:current_task+1]` (`Trainer.py:682-756`); returns the running iteration.  (Joint-encoder mode:
        as in `train_class_incremental`.)"""
        batch_idx = last_batch
        self._set_mode(True)
        cl = CONTINUAL_LEARNING == "myCL" and actual_task is not None and actual_task > 1
        names = self.class_names[:current_task + 1]
        for batch in train_loader:
            if cl:
                self.model_copy()
            batch_idx += 1
            loss = self._train_step(batch, names, criterion, label_cols=slice(0, current_task + 1))
            if cl:
                self.myIncremental(threshold, batch_idx)
            if self.writer is not None:
                self.writer.add_scalar('train/Loss', loss, batch_idx)
        if cl:
            self.myIncremental_save_log(batch_idx)
        return batch_idx

    @torch.no_grad()
    def _eval_loop(self, loader, criterion, epoch, log_tag):
        self._set_mode(False)
        y_true, y_pred, y_score = [], [], []
        batch_idx = 0
        for batch in loader:
            batch_idx += 1
            labels = batch[-1].to(self.device)
            if self._joint is not None:
                if len(batch) < 4:
                    raise ValueError("joint-encoder evaluation needs (images, input_ids, attention_mask, labels) batches")
                new_embs = self.image_model(batch[0].to(self.device))   # (under no_grad: the whole loop is)
            else:
                embs = batch[0].to(self.device)
                new_embs = self.image_adapter(embs) if self._has_img else embs
            logits, loss, cos = self._logits_and_loss(new_embs, labels, self.class_names, criterion, use_grad=False)
            if cos.shape[1] == 2 * len(self.class_names) and TRAIN_LOGIT_DIFF:
                score, pred = K.eval_score(cos.contiguous(), PRED_LOGIT_DIFF)
            else:
                cp, cn = cos[:, 0::2], cos[:, 1::2]
                score = (cp - cn + 2) / 4 if PRED_LOGIT_DIFF else (cp + 1) / 2
                pred = (cp > cn).float()
            if loss is not None and self.writer is not None and log_tag:
                self.writer.add_scalar(log_tag + '/Loss', loss, (epoch - 1) * len(loader) + batch_idx)
            y_true.append(labels.cpu().numpy())
            y_pred.append(pred.cpu().numpy())
            y_score.append(score.cpu().numpy())
        return np.concatenate(y_true), np.concatenate(y_pred), np.concatenate(y_score)

    def val(self, val_loader, criterion, epoch, epochs, mode="joint", tasks_order=None):
        """`Trainer.py:773-866`: scores (pos+1)/2 or (pos-neg+2)/4, predictions argmax([neg,pos]), BCE logged."""
        y_true, y_pred, y_score = self._eval_loop(val_loader, criterion, epoch, "val")
        return self.evaluate_model(y_true, y_pred, y_score, mode, epoch, "val", epochs, tasks_order)

    def test(self, test_loader, criterion, epoch, epochs, mode="joint", tasks_order=None, plot_tsne_array=None):
        """`Trainer.py:989-1062` (the t-SNE / heat-map plots that follow there are out of scope)."""
        y_true, y_pred, y_score = self._eval_loop(test_loader, None, epoch, None)
        return self.evaluate_model(y_true, y_pred, y_score, mode, epoch, "test", epochs, tasks_order)

    @torch.no_grad()
    def evaluate_model(self, y_true, y_pred, y_score, mode, epoch, val_test, epochs, tasks_order):
        """Scalar metrics of `Trainer.py:869-905` (host-side sklearn); returns them as a dict."""
        from sklearn.metrics import accuracy_score, f1_score, roc_auc_score
        m = {"Accuracy": accuracy_score(y_true, y_pred),
             "F1-macro score": f1_score(y_true, y_pred, average="macro", zero_division=0),
             "F1-weighted score": f1_score(y_true, y_pred, average="weighted", zero_division=0)}
        try:
            m["AUROC-macro"] = roc_auc_score(y_true, y_score, average="macro", multi_class="ovr")
            m["AUROC-weighted"] = roc_auc_score(y_true, y_score, average="weighted", multi_class="ovr")
        except ValueError:  # a class without both labels in a tiny synthetic split
            pass
        if self.writer is not None:
            for k, v in m.items():
                self.writer.add_scalar(val_test + "/" + k, v, epoch)
        return m

    # ------------------------------------------------------------------------------------------------ continual learning
    @torch.no_grad()
    def model_copy(self):
        """Snapshot the adapters before a step (`Trainer.py:1634-1641`)."""
        if self._has_img:
            self.image_adapter_copy = [p.detach().clone() for p in self.image_adapter.parameters()]
        if self._has_txt:
            self.text_adapter_copy = [p.detach().clone() for p in self.text_adapter.parameters()]
        if self._joint is not None:   # both encoders: one copy of the flat parameter buffer (0.5 GB)
            self._flat_copy = self.optimizer.flat_p.detach().clone()
        self.n_reset = 0
        self.n_updated = 0
        self._reset_counters.zero_()
        self._reset_total = 0

    @torch.no_grad()
    def _weight_reset(self, threshold):
        for mod, snap in ((self.image_adapter, self.image_adapter_copy), (self.text_adapter, self.text_adapter_copy)):
            if mod is None or snap is None:
                continue
            for p, old in zip(mod.parameters(), snap):
                K.weight_reset(p.data, old, threshold, self._reset_counters)
                self._reset_total += p.numel()
        if self._joint is not None and self._flat_copy is not None:
            # per parameter tensor, as in the reference: each tensor is one gap-free slice of the flat buffer and of its copy
            flat, base = self.optimizer.flat_p, self.optimizer.flat_p.data_ptr()
            for p in self.optimizer.params:
                o, n = (p.data_ptr() - base) // 4, p.numel()
                K.weight_reset(flat[o:o + n], self._flat_copy[o:o + n], threshold, self._reset_counters)
                self._reset_total += n

    @torch.no_grad()
    def myIncremental(self, threshold, iteration):
        """`Trainer.py:1556-1587`: per tensor, restore entries whose |new-old| < min + threshold*(max-min)."""
        self._weight_reset(threshold)

    @torch.no_grad()
    def _reset_stats(self):
        n_reset = int(self._reset_counters[0].item())
        return n_reset, self._reset_total - n_reset

    @torch.no_grad()
    def myIncremental_save_log(self, iteration):
        self.n_reset, self.n_updated = self._reset_stats()
        tot = max(1, self.n_reset + self.n_updated)
        print("\nnumber of resets:", self.n_reset, "number of updates:", self.n_updated, "percentage resets", self.n_reset / tot)
        if self.writer is not None:
            self.writer.add_scalar("monitor-resets/resets", self.n_reset, iteration)
            self.writer.add_scalar("monitor-resets/updates", self.n_updated, iteration)
            self.writer.add_scalar("monitor-resets/percentage resets", self.n_reset / tot, iteration)

    @torch.no_grad()
    def profIncremental(self, epoch, epochs, actual_task, threshold):
        """`Trainer.py:1589-1632`: the same reset applied once per epoch, then counters logged and cleared."""
        self._weight_reset(threshold)
        self.myIncremental_save_log((actual_task - 1) * epochs + epoch)
        self.n_reset = 0
        self.n_updated = 0
        self._reset_counters.zero_()
        self._reset_total = 0

    # ------------------------------------------------------------------------------------------------ checkpoint
    @torch.no_grad()
    def save(self):
        """`Trainer.py:1643-1648`: the adapters into the writer's log dir under the reference's file names.  The reference pickles
        the whole module; here the file holds the module's state dict (tensors only), so that `load` — like every other loader of
        this package — can run with `weights_only=True`."""
        if self._has_img:
            torch.save(self.image_adapter.state_dict(), os.path.join(self.writer.log_dir, 'image_adapter.pt'))
        if self._has_txt:
            torch.save(self.text_adapter.state_dict(), os.path.join(self.writer.log_dir, 'text_adapter.pt'))
        if self._joint is not None and self.rank == 0:
            torch.save(self.image_model.state_dict(), os.path.join(self.writer.log_dir, 'image_model.pt'))
            torch.save(self.bert_encoder.model.state_dict(), os.path.join(self.writer.log_dir, 'text_model.pt'))
        if hasattr(self.writer, "flush"):
            self.writer.flush()

    @torch.no_grad()
    def load(self):
        """Counterpart of `save` (the reference's `load`, `Trainer.py:1650-1655`, mistakenly calls `torch.save` for the text
        adapter).  Files are opened with `weights_only=True`: a pickled module, as the reference writes, is refused."""
        for has, mod, name in ((self._has_img, self.image_adapter, 'image_adapter.pt'), (self._has_txt, self.text_adapter, 'text_adapter.pt')):
            if has:
                path = os.path.join(self.writer.log_dir, name)
                try:
                    sd = torch.load(path, map_location="cpu", weights_only=True)
                except Exception as e:   # a whole-module pickle, as the reference's save() writes (Trainer.py:1646-1648)
                    raise ValueError(f"{path}: not a tensors-only state dict ({type(e).__name__}).  This package stores "
                                     f"`module.state_dict()` under the reference's file names and opens files with weights_only=True; "
                                     f"a file written by the reference's Trainer.save() is a pickled `models.myMLP` module — convert it "
                                     f"in the reference environment (INTEGRATION.md, 'Adapter checkpoints') before loading it here") from e
                if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
                    raise ValueError(f"{path}: expected a state dict of tensors (see INTEGRATION.md, 'Adapter checkpoints')")
                mod.load_state_dict(sd)
        if self._joint is not None:   # the encoders `save` wrote in joint mode; the flat parameter buffer is updated in place
            for mod, name in ((self.image_model, 'image_model.pt'), (self.bert_encoder.model, 'text_model.pt')):
                sd = torch.load(os.path.join(self.writer.log_dir, name), map_location="cpu", weights_only=True)
                with torch.no_grad():
                    own = mod.state_dict()
                    for k, v in sd.items():
                        own[k].copy_(v)
            self._bert_cache.clear()
