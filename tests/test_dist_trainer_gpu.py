"""Data-parallel Trainer with the REAL HIP kernels: 2 ranks share the one GPU of the test box over gloo (RCCL refuses two ranks on
one device; the collectives' semantics are the same).  BASELINE config 4's code path at small size: the class-incremental schedule
(`CLASS_INCREMENTAL.py:67-90`, MORE_LABELS) on the reference's adapter step with every rank training on its row shard of the same
global batch, against ONE process running the global batches: same logged losses, identical replicas, same adapters afterwards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_GLOBAL, N_TRAIN, LR = 64, 640, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_schedule():
    """5 tasks x 2 batches of the class-incremental schedule; returns (losses, flat parameters)"""
    from incremental_multimodal_medical_learning_ii_amd import Trainer as TR
    from incremental_multimodal_medical_learning_ii_amd import synthetic as syn
    from incremental_multimodal_medical_learning_ii_amd.DataRetrieval import CHEXPERT_COMPETITION_CLASSES, create_prompts
    from incremental_multimodal_medical_learning_ii_amd.health_multimodal import text as T
    cfg = T.CXRBertConfig(vocab_size=2048, hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                          max_position_embeddings=32)
    tm = T.CXRBertModel(cfg)
    syn.fill_module_(tm)
    engine = T.TextInferenceEngine(T.SyntheticTokenizer(2048), tm.eval().to("cuda"))
    names = list(CHEXPERT_COMPETITION_CLASSES)
    torch.manual_seed(27)                                        # adapters' default init and the samplers: the same on every rank
    writer = TR.ScalarWriter(os.path.join(os.environ["CXRK_TEST_OUT"], f"w{os.environ.get('RANK', 'single')}"))
    tr = TR.Trainer(False, create_prompts(names), names, "standard", LR, torch.device("cuda"), writer, bert_encoder=engine)
    train, _, _ = TR.Trainer.synthetic_loaders(N_TRAIN, 64, 64, B_GLOBAL)
    tasks = TR.Trainer.split_dataloader_data_incremental(TR.Trainer.concat_to_tensor_dataloader(train), 5)
    crit = torch.nn.BCEWithLogitsLoss()
    last = 0
    for t, loader in enumerate(tasks):
        last = tr.train_class_more_labels_incremental(loader, crit, 1, None, None, t, last, t + 1)
    torch.cuda.synchronize()
    return [v for _, v, _ in writer.scalars("train/Loss")], tr.optimizer.flat_p.detach().cpu().numpy(), tr


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), CXRK_TEST_OUT=out_dir)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    losses, flat, tr = _run_schedule()
    assert tr.world == world and tr.rank == rank
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), losses=np.array(losses), flat=flat)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_class_incremental_adapter_schedule_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    os.environ["CXRK_TEST_OUT"] = str(tmp_path)
    os.environ.pop("RANK", None)
    losses, flat, tr = _run_schedule()
    assert tr.world == 1 and len(losses) == 10
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    np.testing.assert_array_equal(r[0]["flat"], r[1]["flat"])                                 # replicas stay identical
    for k in range(world):
        np.testing.assert_allclose(r[k]["losses"], losses, rtol=1e-4)                         # global-batch loss on every rank
    # 10 Adam steps at lr 1e-3 move a weight by up to ~1e-2; the sharded sum differs from the one-pass sum in the last bits only
    assert float(np.abs(r[0]["flat"] - flat).max()) < 5e-5
