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


def create_reference_embedding2(model, dataloader, device, cache=True):
    """oc_classifier.py:159-202: mean embedding of the bona-fide set, threshold = largest distance to it."""
    if cache and os.path.exists("reference_embedding.pt") and os.path.exists("threshold.pt"):
        print("Loading reference embedding and threshold...")
        return torch.load("reference_embedding.pt"), torch.load("threshold.pt")
    print("Creating a reference embedding...")
    model.eval()
    embs = []
    with torch.no_grad():
        for data, _ in dataloader:
            emb, _ = model(data.to(device))
            embs.append(emb.clone())
    embs = torch.cat(embs, dim=0)                                   # [N,160]
    reference_embedding = embs.mean(dim=0, keepdim=True)            # [1,160] like torch.mean(torch.stack(..), 0)
    dist = ops.pairwise_dist(reference_embedding.reshape(-1).contiguous(), embs.contiguous())
    with open("distances.txt", "a") as f:
        for d in dist.tolist():
            f.write(f"{d}\n")
    threshold = dist.max()
    if cache:
        torch.save(reference_embedding, "reference_embedding.pt")
        torch.save(threshold, "threshold.pt")
    return reference_embedding, threshold


def score_eval_set_1c2(model, dataloader, device, reference_embedding, threshold, path="scores.txt"):
    """One-class scoring (oc_classifier.py:243-265): distance to the reference embedding, 1 when above the threshold."""
    model.eval()
    thr = float(threshold)
    print("Scoring the evaluation set...")
    with open(path, "w") as f, torch.no_grad():
        for idx, (data, _) in enumerate(dataloader):
            emb, _ = model(data.to(device))
            d = float(ops.pairwise_dist(reference_embedding.reshape(-1).contiguous(), emb.contiguous())[0])
            f.write(f"{d}, 1 \n" if d > thr else f"{d}, 0 \n")


def score_eval_set_2c2(model, dataloader, device, path="scores.txt"):
    """Two-class scoring (oc_classifier.py:292-312): the bona-fide logit."""
    model.eval()
    with open(path, "w") as f, torch.no_grad():
        for data, _ in dataloader:
            _, out = model(data.to(device))
            f.write(f"{float(out[0][0])}\n")


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
    args = parser.parse_args(argv)
    from .models.sslassist import AModel
    device = torch.device("cuda")
    # The reference builds AModel (which loads the fairseq file from a hard-coded path) and then overwrites every tensor with the trained
    # checkpoint (oc_classifier.py:335-340, strict).  That checkpoint already holds all ``ssl_model.model.*`` tensors, so here it is the one
    # source of the XLS-R weights; a tensor that is missing or has the wrong shape is an error, never a silently random front-end.
    sd = torch.load(args.pretrained_sslaasist, map_location="cpu")
    ssl = {k[len("ssl_model.model."):]: v for k, v in sd.items() if k.startswith("ssl_model.model.")}
    aasist = AModel(None, device, ssl_state_dict=ssl)
    aasist.load_state_dict(sd, strict=True)
    print("Pretrained weights loaded")
    train_loader = DataLoader(ASVDataset(args.protocol_file, args.dataset_dir), batch_size=1, shuffle=False, num_workers=0)
    reference_embedding, threshold = create_reference_embedding2(aasist, train_loader, device)
    eval_loader = DataLoader(ASVDataset(args.eval_protocol_file, args.eval_dataset_dir, eval=True), batch_size=1, shuffle=False, num_workers=0)
    if args.two_class:
        score_eval_set_2c2(aasist, eval_loader, device)
    else:
        score_eval_set_1c2(aasist, eval_loader, device, reference_embedding, threshold)
    print(f"threshold = {threshold}")


if __name__ == "__main__":
    main()
