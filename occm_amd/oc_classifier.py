"""Scoring entry point -- mirror of the reference's ``oc_classifier.py`` (ASVDataset :27-110, reference embedding
:159-202, scoring :243-312, CLI :315-361).

    python -m occm_amd.oc_classifier --pretrained-sslaasist CKPT --protocol_file P --dataset_dir D \
                                     --eval_protocol_file EP --eval_dataset_dir ED

Writes ``reference_embedding.pt``, ``threshold.pt``, ``distances.txt`` and ``scores.txt`` (``"{dist}, {0|1} \\n"``, the
format evaluate.py:85-101 parses).  Distances come from the occ_pairwise_dist kernel.
"""
import argparse
import os

import torch
from torch.utils.data import DataLoader, Dataset

from . import ops
from .data_utils_SSL import load_audio


SSL_DTYPES = {"f32": torch.float32, "f32x3": torch.float32, "bf16": torch.bfloat16}


class ASVDataset(Dataset):
    """eval=False: bona-fide rows of a 5-column protocol (for the reference embedding); eval=True: every row of a trial
    list whose first column is the utterance id (oc_classifier.py:58-78)."""

    def __init__(self, protocol_file, dataset_dir, eval=False, loader=load_audio):
        self.protocol_file, self.dataset_dir, self.eval = protocol_file, dataset_dir, eval
        self.file_list, self.label_list = [], []
        self._load = loader
        with open(protocol_file, "r") as f:
            for line in f:
                parts = line.strip().split(" ")
                if eval:
                    self.file_list.append(parts[0]); self.label_list.append("unknown")
                elif parts[4] == "bonafide":
                    self.file_list.append(parts[1]); self.label_list.append(parts[4])
        self._length = len(self.file_list)

    def __len__(self):
        return self._length

    def __getitem__(self, idx):
        name = self.file_list[idx]
        path = os.path.join(self.dataset_dir, name + ".flac")
        if not os.path.exists(path):
            path = os.path.join(self.dataset_dir, name + ".wav")
        feature, _ = self._load(path)
        label = [1 if self.label_list[idx] == "spoof" else 0]
        return torch.tensor(feature, dtype=torch.float32), torch.tensor(label, dtype=torch.int64)


def n_frames(L):
    """Frames the wav2vec2 conv stack makes of L samples (kernels 10,3,3,3,3,2,2; strides 5,2,2,2,2,2,2)."""
    for k, s in ((10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)):
        L = (L - k) // s + 1
    return L


def canonical_len(T):
    """Samples that T frames depend on: receptive field 400, hop 320.  The conv stack has no padding, so samples past this point
    complete no further frame and change nothing: cropping an utterance to canonical_len(n_frames(L)) leaves its features unchanged."""
    return (T - 1) * 320 + 400


PAD_FRAMES = 16          # masked batches: utterances whose frame counts fall into the same 16-frame band share a batch (<= 8 % padding at 4 s)


def embed_dataset(model, dataloader, device, batch_size=1, rank=0, world=1, masked=None):
    """(emb [N,160] (AASIST) or [N,128] (SE-ResNet34), logits [N,2]) of every utterance, in dataset order.

    world > 1 (one process per GPU, torch.distributed initialised): rank r embeds the utterances whose position p has p % world == r
    -- the loader must then be built over ``shard_dataset(dataset, rank, world)`` -- and a sum all-reduce of the zero-initialised
    result tensors gives every rank the whole set ("shard the file list, all-gather the embeddings").

    batch_size 1 is the reference's loop (oc_classifier.py:182-186, 243-265: one utterance per forward).  batch_size > 1 batches
    utterances of SIMILAR length: those whose frame counts fall into one band of PAD_FRAMES frames are cropped to the
    samples their frames depend on, zero-padded to the longest of the batch and run through the front-end together with key masks
    (``model(x, lengths=...)``: the positional conv sees zeros past each utterance's end and attention ignores the pad keys, as in the
    un-padded run); the back-end runs per distinct frame count.  Per-utterance results differ from the one-at-a-time loop only through
    the GEMM kernels picked for a different row count (summation order).  masked=False (or a model without ``supports_lengths``) falls
    back to batches of EQUAL frame count, which need no mask at all.  Buckets fill as the loader delivers utterances; at most
    (#bands in flight) x batch_size waveforms are held."""
    model.eval()
    ds = dataloader.dataset
    n_local = len(ds)
    n = getattr(ds, "full_len", n_local)                     # shard_dataset records the length of the whole set
    width = int(getattr(model, "emb_dim", 160))              # AASIST: 160 (5 x 32 readouts); SE-ResNet34: 128
    embs, logits = torch.zeros(n, width, device=device), torch.zeros(n, 2, device=device)
    pending = {}
    if masked is None:
        masked = bool(getattr(model, "supports_lengths", False))
    masked = masked and batch_size > 1

    def flush(items):
        idxs = torch.tensor([i * world + rank if world > 1 else i for i, _ in items], device=device)
        lens = [w.numel() for _, w in items]
        if masked and min(lens) != max(lens):
            x = torch.zeros(len(items), max(lens))
            for j, (_, w) in enumerate(items):
                x[j, : w.numel()] = w
            emb, out = model(x.to(device), lengths=lens)
        else:
            emb, out = model(torch.stack([w for _, w in items]).to(device))
        embs[idxs] = emb.float(); logits[idxs] = out.float()

    pos = 0
    with torch.no_grad():
        for data, _ in dataloader:
            for row in data:                                   # the loader may itself deliver more than one utterance
                if batch_size <= 1:
                    flush([(pos, row)])
                else:
                    T = n_frames(row.numel())
                    if T < 1:
                        raise ValueError("utterance %d: %d samples are too short for the conv stack" % (pos, row.numel()))
                    key = (T - 1) // PAD_FRAMES if masked else T
                    bucket = pending.setdefault(key, [])
                    bucket.append((pos, row[:canonical_len(T)]))
                    if len(bucket) == batch_size:
                        flush(pending.pop(key))
                pos += 1
        left = [it for key in sorted(pending) for it in pending[key]]
        if masked:                                             # the partly filled bands, shortest first, in batches of neighbours
            left.sort(key=lambda it: it[1].numel())
            for i in range(0, len(left), batch_size):
                flush(left[i:i + batch_size])
        else:
            for key in sorted(pending):
                flush(pending[key])
    if pos != n_local:
        raise RuntimeError("loader delivered %d of %d utterances" % (pos, n_local))
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(embs); dist.all_reduce(logits)
    return embs, logits


class _Shard(torch.utils.data.Dataset):
    def __init__(self, base, rank, world):
        self.base, self.rank, self.world, self.full_len = base, rank, world, len(base)

    def __len__(self):
        return (self.full_len - self.rank + self.world - 1) // self.world

    def __getitem__(self, i):
        return self.base[i * self.world + self.rank]


def shard_dataset(dataset, rank, world):
    """Utterances rank, rank + world, ... of the set (the whole set on one rank)."""
    return dataset if world <= 1 else _Shard(dataset, rank, world)


class ExtractorEncoder:
    """The two-model form of the reference's scoring functions (oc_classifier.py:139-144, 229-234, 281-287):
    ``emb = extractor(data); emb = emb.unsqueeze(1); encoder(emb)`` -- an SSLModel feature extractor followed by se_resnet34 -- behind the
    one-model call signature embed_dataset drives."""
    emb_dim = 128

    def __init__(self, extractor, encoder):
        self.extractor, self.encoder = extractor, encoder

    def eval(self):
        self.extractor.eval(); self.encoder.eval()
        return self

    supports_lengths = True

    def __call__(self, data, lengths=None):
        if lengths is None:
            feats = self.extractor(data)
            return self.encoder(feats.float().unsqueeze(1))
        # zero-padded batch of unequal utterances: masked front-end, the encoder once per distinct frame count (its pooling depends on it)
        feats = self.extractor(data, lengths=lengths).float()
        fr = [n_frames(int(v)) for v in lengths]
        com = torch.empty(len(fr), self.emb_dim, device=feats.device); des = torch.empty(len(fr), 2, device=feats.device)
        for T in sorted(set(fr)):
            idx = torch.tensor([i for i, t in enumerate(fr) if t == T], device=feats.device)
            c, d = self.encoder(feats[idx, :T].contiguous().unsqueeze(1))
            com[idx] = c; des[idx] = d
        return com, des


def create_reference_embedding(extractor, encoder, dataloader, device, cache=True, batch_size=1, rank=0, world=1):
    """oc_classifier.py:113-157: as create_reference_embedding2 with the (extractor, encoder) pair."""
    return create_reference_embedding2(ExtractorEncoder(extractor, encoder), dataloader, device, cache, batch_size, rank, world, distances_file=None)


def score_eval_set_1c1(extractor, encoder, dataloader, device, reference_embedding, threshold, path="scores.txt", batch_size=1, rank=0, world=1):
    """oc_classifier.py:206-241: one-class scoring with the (extractor, encoder) pair."""
    return score_eval_set_1c2(ExtractorEncoder(extractor, encoder), dataloader, device, reference_embedding, threshold, path, batch_size, rank, world)


def score_eval_set_2c1(extractor, encoder, dataloader, device, path="scores.txt", batch_size=1, rank=0, world=1):
    """oc_classifier.py:268-291: two-class scoring (the bona-fide logit) with the (extractor, encoder) pair."""
    return score_eval_set_2c2(ExtractorEncoder(extractor, encoder), dataloader, device, path, batch_size, rank, world)


def create_reference_embedding2(model, dataloader, device, cache=True, batch_size=1, rank=0, world=1, distances_file="distances.txt"):
    """oc_classifier.py:159-202: mean embedding of the bona-fide set, threshold = largest distance to it.  (The reference's AASIST form
    also appends every distance to distances.txt, :197-199; its two-model form :113-157 does not.)"""
    if cache and os.path.exists("reference_embedding.pt") and os.path.exists("threshold.pt"):
        print("Loading reference embedding and threshold...")
        return torch.load("reference_embedding.pt"), torch.load("threshold.pt")
    print("Creating a reference embedding...")
    embs, _ = embed_dataset(model, dataloader, device, batch_size, rank, world)      # [N,160]
    reference_embedding = embs.mean(dim=0, keepdim=True)            # [1,160] like torch.mean(torch.stack(..), 0)
    dist = ops.pairwise_dist(reference_embedding.reshape(-1).contiguous(), embs.contiguous())
    if rank == 0 and distances_file:
        with open(distances_file, "a") as f:
            for d in dist.tolist():
                f.write(f"{d}\n")
    threshold = dist.max()
    if cache and rank == 0:
        torch.save(reference_embedding, "reference_embedding.pt")
        torch.save(threshold, "threshold.pt")
    return reference_embedding, threshold


def score_eval_set_1c2(model, dataloader, device, reference_embedding, threshold, path="scores.txt", batch_size=1, rank=0, world=1):
    """One-class scoring (oc_classifier.py:243-265): distance to the reference embedding, 1 when above the threshold."""
    thr = float(threshold)
    print("Scoring the evaluation set...")
    embs, _ = embed_dataset(model, dataloader, device, batch_size, rank, world)
    dist = ops.pairwise_dist(reference_embedding.reshape(-1).contiguous(), embs.contiguous()).tolist()
    if rank != 0:
        return
    with open(path, "w") as f:
        for d in dist:
            f.write(f"{d}, 1 \n" if d > thr else f"{d}, 0 \n")


def score_eval_set_2c2(model, dataloader, device, path="scores.txt", batch_size=1, rank=0, world=1):
    """Two-class scoring (oc_classifier.py:292-312): the bona-fide logit."""
    _, logits = embed_dataset(model, dataloader, device, batch_size, rank, world)
    if rank != 0:
        return
    with open(path, "w") as f:
        for v in logits[:, 0].tolist():
            f.write(f"{v}\n")


def main(argv=None):
    parser = argparse.ArgumentParser(description="One-class classifier")
    parser.add_argument("--pretrained-sslaasist", type=str, default="/datac/longnv/occm/aasist_vocoded_1.pt")
    parser.add_argument("--pretrained-ssl", type=str, default="/datac/longnv/occm/ssl_triplet_1.pt")
    parser.add_argument("--pretrained-senet", type=str, default="/datac/longnv/occm/senet34_triplet_1.pt")
    parser.add_argument("--protocol_file", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof_LA_cm_protocols/ASVspoof2019.LA.cm.train.trn.txt")
    parser.add_argument("--dataset_dir", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_train/flac")
    parser.add_argument("--eval_protocol_file", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof_LA_cm_protocols/ASVspoof2019.LA.cm.eval.trl.txt")
    parser.add_argument("--eval_dataset_dir", type=str, default="/datab/Dataset/ASVspoof/LA/ASVspoof2019_LA_eval/flac")
    parser.add_argument("--two_class", action="store_true", help="score with the bona-fide logit (score_eval_set_2c2)")
    parser.add_argument("--batch_size", type=int, default=1, help="1 = the reference's one-utterance loop; > 1 = batches of utterances of similar length, zero-padded "
                        "to the longest, with key masks in the front-end (each utterance's result equals its one-at-a-time result)")
    parser.add_argument("--num_workers", type=int, default=0)
    parser.add_argument("--ssl_dtype", choices=["f32", "f32x3", "bf16"], default="f32",
                        help="arithmetic of the XLS-R front-end and the AASIST GEMMs: f32 (default) = exact-f32 MFMA, the path that meets the 1e-3 "
                             "parity bar against the reference's fp32 (embeddings within 2e-5 of the CPU oracle on the 24-layer model); bf16 = bf16 "
                             "MFMA with f32 accumulate, the training arithmetic: faster, but on the 24-layer model embeddings move by up to ~2.5 and "
                             "distances by up to ~1 (DESIGN.md section 5, profiles/r03_scoring_parity_300m.json); f32x3 = f32 storage, every GEMM operand "
                             "split into bf16 hi + lo and multiplied with three bf16 MFMAs (products exact to 2^-16): the f32 path's accuracy class at a "
                             "multiple of its GEMM rate")
    args = parser.parse_args(argv)
    from .models.sslassist import AModel
    from . import parallel
    rank, world, local = parallel.init_from_env()              # torchrun --nproc-per-node N: the file lists are sharded over the ranks
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    device = torch.device("cuda")
    # The reference builds AModel (which loads the fairseq file from a hard-coded path) and then overwrites every tensor with the trained
    # checkpoint (oc_classifier.py:335-340, strict).  That checkpoint already holds all ``ssl_model.model.*`` tensors, so here it is the one
    # source of the XLS-R weights; a tensor that is missing or has the wrong shape is an error, never a silently random front-end.
    sd = torch.load(args.pretrained_sslaasist, map_location="cpu")
    ssl = {k[len("ssl_model.model."):]: v for k, v in sd.items() if k.startswith("ssl_model.model.")}
    aasist = AModel(None, device, ssl_state_dict=ssl, ssl_dtype=SSL_DTYPES[args.ssl_dtype],      # back-end compute follows (f32 -> "f32")
                    ssl_f32_gemm="x3" if args.ssl_dtype == "f32x3" else "exact")
    aasist.load_state_dict(sd, strict=True)
    print("Pretrained weights loaded")
    kw = dict(batch_size=args.batch_size, rank=rank, world=world)
    train_loader = DataLoader(shard_dataset(ASVDataset(args.protocol_file, args.dataset_dir), rank, world), batch_size=1, shuffle=False, num_workers=args.num_workers)
    reference_embedding, threshold = create_reference_embedding2(aasist, train_loader, device, **kw)
    eval_loader = DataLoader(shard_dataset(ASVDataset(args.eval_protocol_file, args.eval_dataset_dir, eval=True), rank, world), batch_size=1, shuffle=False,
                             num_workers=args.num_workers)
    if args.two_class:
        score_eval_set_2c2(aasist, eval_loader, device, **kw)
    else:
        score_eval_set_1c2(aasist, eval_loader, device, reference_embedding, threshold, **kw)
    print(f"threshold = {threshold}")


if __name__ == "__main__":
    main()
