"""Training entry point with the reference's surface (reference train.py:14-32):

    python -m music_generator_amd.train            # one GPU
    torchrun --nproc-per-node 8 -m music_generator_amd.train      # data parallel over xGMI

Extra flags (the reference's train.py defines none): --batch-size / --epochs / --time-steps / --dtype /
--synthetic N (train on N synthetic windows when no MIDI corpus is present).  BASELINE configs[0] (the
plumbing run on 4 synthetic MIDI files) is `--batch-size 2 --time-steps 8 --epochs 1`."""
import argparse
import os

from .callbacks import EarlyStopping, ModelCheckpoint, TensorBoard
from .constants import *  # noqa: F401,F403
from .dataset import load_all
from .util import build_or_load


def _init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm


def train(models, batch_size=BATCH_SIZE, epochs=1000, data=None, time_steps=SEQ_LEN):
    """reference train.py:18-29"""
    print('Loading data')
    train_data, train_labels = data if data is not None else load_all(styles, batch_size, time_steps)
    cbs = [
        ModelCheckpoint(MODEL_FILE, monitor='loss', save_best_only=True, save_weights_only=True),
        EarlyStopping(monitor='loss', patience=5),
        TensorBoard(log_dir='out/logs', histogram_freq=1),
    ]
    print('Training')
    return models[0].fit(train_data, train_labels, epochs=epochs, callbacks=cbs, batch_size=batch_size)


def main(argv=None):
    ap = argparse.ArgumentParser(description='Trains the DeepJ model on MI355X.')
    ap.add_argument('--batch-size', type=int, default=BATCH_SIZE, help='global batch (split over ranks)')
    ap.add_argument('--epochs', type=int, default=1000)
    ap.add_argument('--time-steps', type=int, default=SEQ_LEN, help='window length (constants.py SEQ_LEN)')
    ap.add_argument('--dtype', default=None, choices=['f32', 'bf16'])
    ap.add_argument('--synthetic', type=int, default=0, help='train on N synthetic windows instead of data/')
    args = ap.parse_args(argv)
    _init_distributed()
    models = build_or_load(time_steps=args.time_steps, dtype=args.dtype)
    data = None
    if args.synthetic:
        from .data import synthetic_batch
        n, c, b, s, t = synthetic_batch(NUM_NOTES, args.time_steps, args.synthetic, seed=0)
        data = ([n, c, b, s], [t])
    return train(models, args.batch_size, args.epochs, data, args.time_steps)


if __name__ == '__main__':
    main()
