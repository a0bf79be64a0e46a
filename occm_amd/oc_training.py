"""Training entry point -- mirror of the reference's ``oc_training.py`` (PFDataset :31-270, loop :273-401).

    python -m occm_amd.oc_training --train_dataset_dir DIR --train_protocol_file FILE [...]

Same CLI flag names (:280-289) plus the knobs the reference hard-codes (lr :324, epochs :342, loss weights
:380-381, vocoded dir :72, RawBoost algo :221).  Outputs match: ``aasist_vocoded_{epoch}.pt`` (state_dict with the
reference key names) and ``loss.txt``.  One process per GPU (torchrun) instead of nn.DataParallel; wandb is optional.
"""
import argparse
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .data_utils_SSL import load_audio, process_Rawboost_feature

VOCODERS = ["hifigan", "hn-sinc-nsf-hifi", "hn-sinc-nsf", "melgan", "waveglow"]          # oc_training.py:174


def rawboost_args(argv=()):
    """The 17 RawBoost flags with the defaults of oc_training.py:79-119."""
    p = argparse.ArgumentParser(add_help=False)
    p.add_argument("--algo", type=int, default=3)
    for name, d in (("nBands", 5), ("minF", 20), ("maxF", 8000), ("minBW", 100), ("maxBW", 1000), ("minCoeff", 10), ("maxCoeff", 100),
                    ("minG", 0), ("maxG", 0), ("minBiasLinNonLin", 5), ("maxBiasLinNonLin", 20), ("N_f", 5), ("P", 10), ("g_sd", 2),
                    ("SNRmin", 10), ("SNRmax", 40)):
        p.add_argument("--" + name, type=int, default=d)
    return p.parse_known_args(list(argv))[0]


class PFDataset(Dataset):
    """Groups of 12 utterances per item: 6 bona fide (the indexed one + 5 random), 1 spoof, 5 vocoded copies of the
    indexed bona fide; zero-padded to the longest; labels spoof=1 / bonafide=0 (oc_training.py:201-256)."""

    def __init__(self, protocol_file, dataset_dir, vocoded_dir="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_vocoded", rawboost_algo=0,
                 loader=load_audio):
        self.protocol_file, self.dataset_dir = protocol_file, dataset_dir
        self.file_list, self.label_list = [], []
        with open(protocol_file, "r") as f:
            for line in f:
                parts = line.strip().split(" ")
                self.file_list.append(parts[1])
                self.label_list.append(parts[4])
        self.spoof_indices = [i for i, l in enumerate(self.label_list) if l == "spoof"]
        self.bonafide_indices = [i for i, l in enumerate(self.label_list) if l == "bonafide"]
        self._length = len(self.bonafide_indices)
        self._vocoded_dir = vocoded_dir
        self.rawboost_algo = rawboost_algo
        self.args = rawboost_args()
        self._load = loader

    def _get_random_files(self, indices_list, exclude_idx, number_needed):
        possible = list(set(indices_list) - {exclude_idx}) if exclude_idx is not None else list(indices_list)
        if len(possible) < number_needed:
            raise ValueError("Not enough files to select from.")                           # oc_training.py:147-148
        return [self.file_list[i] for i in random.sample(possible, k=number_needed)]

    def _get_vocoded_files(self, bonafide):
        return [f"{v}_{bonafide}" for v in VOCODERS]

    def _get_files(self, idx):
        bona = self._get_random_files(self.bonafide_indices, idx, 5)
        spoof = self._get_random_files(self.spoof_indices, None, 1)
        d = {"bona1": self.file_list[idx], "spoof1": spoof[0]}
        for i, f in enumerate(bona):
            d["bona%d" % (i + 2)] = f
        return d

    def __len__(self):
        return self._length

    def __getitem__(self, idx):
        files = self._get_files(idx)
        feats, labels = [], []
        for key, name in sorted(files.items()):
            x, sr = self._load(os.path.join(self.dataset_dir, name + ".wav"))
            if self.rawboost_algo:
                x = np.asarray(process_Rawboost_feature(x, sr, self.args, self.rawboost_algo), dtype=np.float32)
            feats.append(x)
            labels.append(1 if key.startswith("spoof") else 0)
        for name in self._get_vocoded_files(files["bona1"]):
            x, sr = self._load(os.path.join(self._vocoded_dir, name + ".wav"))
            if self.rawboost_algo:
                x = np.asarray(process_Rawboost_feature(x, sr, self.args, self.rawboost_algo), dtype=np.float32)
            feats.append(x)
            labels.append(1)
        max_len = max(f.shape[0] for f in feats)
        padded = np.stack([np.pad(f, (0, max_len - len(f)), mode="constant") for f in feats])
        return torch.tensor(padded, dtype=torch.float32), torch.tensor(np.array(labels), dtype=torch.int64)


def main(argv=None):
    parser = argparse.ArgumentParser(description="Train a model on a dataset")
    parser.add_argument("--train_dataset_dir", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_train/wav")
    parser.add_argument("--test_dataset_dir", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_eval/flac")
    parser.add_argument("--model", type=str, default="ssl_resnet34")
    parser.add_argument("--finetuned", action="store_true", default=False)
    parser.add_argument("--train_protocol_file", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof_LA_cm_protocols/ASVspoof2019.LA.cm.train.trn.txt")
    parser.add_argument("--test_protocol_file", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof_LA_cm_protocols/ASVspoof2019.LA.cm.eval.trl.txt")
    # literals of the reference lifted to flags (defaults = the committed values)
    parser.add_argument("--lr", type=float, default=1e-5)
    parser.add_argument("--epochs", type=int, default=100)
    parser.add_argument("--w_compact", type=float, default=0.0)
    parser.add_argument("--w_descr", type=float, default=1.0)
    parser.add_argument("--vocoded_dir", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_vocoded")
    parser.add_argument("--rawboost_algo", type=int, default=0)
    parser.add_argument("--ssl_checkpoint", type=str, default="/datac/longnv/SSL_Anti-spoofing/pretrained/xlsr2_300m.pt",
                        help="fairseq XLS-R checkpoint ({'model': ..., 'cfg': ...} or a bare state dict); default = the path the reference hard-codes "
                        "(sslassist.py:24).  A missing file is an error, as in the reference")
    parser.add_argument("--ssl_dtype", choices=["bf16", "f32"], default="bf16",
                        help="arithmetic of a FROZEN XLS-R front-end and of the back-end GEMMs: bf16 MFMA with f32 accumulate (default, the throughput path) "
                             "or f32 = exact-f32 MFMA (the 1e-3 parity path, ~8x slower).  --finetuned always trains with bf16 operands over f32 master weights")
    parser.add_argument("--synthetic_ssl", action="store_true", help="tests only: deterministic random XLS-R weights instead of a checkpoint")
    for _k in ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "dropout_input"):
        parser.add_argument("--ssl_" + _k, type=float, default=None, help="fairseq train-mode %s of XLS-R when --finetuned (default: the checkpoint's cfg)" % _k)
    parser.add_argument("--wandb", action="store_true")
    # not in the reference: the back-end class (its --model flag is parsed but never read: oc_training.py:320 always builds AModel),
    # loader workers (the reference uses num_workers=0; at thousands of utterances/s the decode has to run ahead of the GPU)
    parser.add_argument("--backend", type=str, default="aasist", choices=["aasist", "senet"])
    parser.add_argument("--num_workers", type=int, default=0)
    parser.add_argument("--rawboost_on_gpu", action="store_true", help="apply --rawboost_algo to the whole group in one batched device call inside the "
                        "training step (Philox parameter draws) instead of per utterance in the dataset (np.random draws in the reference's order); "
                        "the batched form also perturbs the zero-padded tails of the shorter utterances of a group")
    args = parser.parse_args(argv)
    if args.num_workers > 0 and args.rawboost_algo and not args.rawboost_on_gpu:
        raise ValueError("dataset-side RawBoost runs on the GPU of the main process: use --num_workers 0 or --rawboost_on_gpu")
    ssl_kw = {"synthetic_ssl": True} if args.synthetic_ssl else {"ssl_cp_path": args.ssl_checkpoint}
    if args.finetuned and args.ssl_dtype != "bf16":
        raise ValueError("--finetuned trains XLS-R with bf16 MFMA operands over f32 master weights; --ssl_dtype f32 applies to a frozen front-end only")
    ssl_kw["ssl_dtype"] = torch.float32 if args.ssl_dtype == "f32" else torch.bfloat16
    if not args.synthetic_ssl and not os.path.exists(args.ssl_checkpoint):
        raise FileNotFoundError("XLS-R checkpoint %s not found (--ssl_checkpoint); the reference loads it unconditionally (sslassist.py:24-26)" % args.ssl_checkpoint)
    print("*************************************************")
    for k in ("train_dataset_dir", "test_dataset_dir", "model", "finetuned", "train_protocol_file", "test_protocol_file"):
        print(f"{k} = {getattr(args, k)}")
    print("*************************************************")

    from . import parallel
    from .models.sslassist import AModel
    from .trainer import OcTrainer
    rank, world, local = parallel.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(local)
    dataset = PFDataset(args.train_protocol_file, args.train_dataset_dir, vocoded_dir=args.vocoded_dir,
                        rawboost_algo=0 if args.rawboost_on_gpu else args.rawboost_algo)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
    loader = DataLoader(dataset, batch_size=1, shuffle=sampler is None, sampler=sampler, num_workers=args.num_workers, pin_memory=True,
                        persistent_workers=args.num_workers > 0, prefetch_factor=4 if args.num_workers > 0 else None)
    # --finetuned: train XLS-R end to end as the reference's optimizer does (oc_training.py:324 holds every SSL parameter); without it
    # the front-end is frozen (BASELINE configs[1]) and its features for the next batch are computed under the current update
    ft = "full" if args.finetuned else False
    if args.backend == "senet":
        from .models.senet import ssl_resnet34
        model = ssl_resnet34(device, finetune_ssl=ft, **ssl_kw)
    else:
        model = AModel(None, device, finetune_ssl=ft, **ssl_kw)
    tc = model.ssl_model.train_cfg
    for _k in ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop", "dropout_input"):
        if getattr(args, "ssl_" + _k) is not None:
            setattr(tc, _k, getattr(args, "ssl_" + _k))
    model.train()
    trainer = OcTrainer(model, lr=args.lr, w_compact=args.w_compact, w_descr=args.w_descr, train_frontend=bool(ft),
                        rawboost_algo=args.rawboost_algo if args.rawboost_on_gpu else 0, rank=rank)
    wb = None
    if args.wandb and rank == 0:
        import wandb as wb
        wb.init(project="oc_classifier")
    for epoch in range(args.epochs):
        print(f"Epoch {epoch + 1}\n-------------------------------")
        if sampler is not None:
            sampler.set_epoch(epoch)
        run_c = torch.zeros(1, device=device); run_d = torch.zeros(1, device=device)
        # one batch of look-ahead: the trainer runs the frozen front-end of batch i+1 on a side stream under batch i's back-end update
        batches = ((inp.squeeze(0).to(device, non_blocking=True), lab.squeeze(0).to(device, non_blocking=True)) for inp, lab in loader)
        cur = next(batches, None)
        i = -1
        while cur is not None:
            i += 1
            nxt = next(batches, None)
            lc, ld = trainer.step(cur[0], cur[1], next_wav=None if nxt is None else nxt[0])
            cur = nxt
            run_c += lc; run_d += ld                                    # stays on the GPU: no per-step host sync
            if i % 100 == 99 and rank == 0:
                c, d = float(run_c) / (i + 1), float(run_d) / (i + 1)
                print(f"[{epoch + 1}, {i + 1}] Train Loss: {args.w_compact * c + args.w_descr * d:.3f}")
                with open("loss.txt", "a") as f:
                    f.write(f"epoch = {epoch + 1}, i = {i + 1}, loss = {args.w_compact * c + args.w_descr * d:.3f}, "
                            f"closs = {args.w_compact * c:.3f}, dloss = {args.w_descr * d:.3f} \n")
                if wb is not None:
                    wb.log({"Epoch": epoch, "Train Compactness Loss": c, "Train Descriptiveness Loss": d})
        if rank == 0:
            print("Saving the models...")
            torch.save(model.state_dict(), f"aasist_vocoded_{epoch}.pt")


if __name__ == "__main__":
    main()
