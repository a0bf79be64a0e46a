"""``calculate_eer.py`` of the reference (:1-38): merge a 5-column protocol with a ``utt score`` file, print the EER.

    python -m occm_amd.calculate_eer --eval_protocol_file P --score_file S
Unlike the reference the argument parser only runs under ``__main__`` so the function is importable.
"""
import argparse

import numpy as np

from .evaluate_metrics import compute_eer


def calculate_EER(eval_protocol_file, score_file, verbose=True):
    labels = {}
    with open(eval_protocol_file) as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 5:
                labels[parts[1]] = parts[4]              # columns: sid utt phy attack label (calculate_eer.py:13)
    spoof, bona = [], []
    with open(score_file) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 2 or parts[0] not in labels:
                continue
            lab = labels[parts[0]]                          # the reference selects exactly these two label strings (:21-22)
            if lab == "spoof":
                spoof.append(float(parts[1]))
            elif lab == "bonafide":
                bona.append(float(parts[1]))
    eer, threshold = compute_eer(np.array(bona), np.array(spoof))   # calculate_eer.py:25 argument order
    if verbose:
        print(f"EER = {eer*100.0}, threshold = {threshold}")
    return eer, threshold


if __name__ == "__main__":
    argparser = argparse.ArgumentParser(description="EER from a protocol and a score file")
    argparser.add_argument("--eval_protocol_file", type=str, default="./database/protocols/PartialSpoof_LA_cm_protocols/PartialSpoof.LA.cm.eval.trl.txt")
    argparser.add_argument("--score_file", type=str, default="./se_resnet34_eval_scores.txt")
    a = argparser.parse_args()
    print(f"eval_protocol_file = {a.eval_protocol_file}")
    print(f"score_file = {a.score_file}")
    calculate_EER(a.eval_protocol_file, a.score_file)
