"""Host-side metrics of the reference (src/postprocess.py): character
accuracy, word-level edit-distance error, attention images, EOS trimming.
The reference uses the `editdistance` package; a small Levenshtein routine
replaces it here."""
import numpy as np
import torch


def trim_eos(sequence):
    """Keep everything up to and including the first '>' (index 1)."""
    out = []
    for ch in sequence:
        out.append(int(ch))
        if int(ch) == 1:
            break
    return out


def edit_distance(a, b):
    """Levenshtein distance between two sequences of hashables."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def calc_acc(predict, label):
    """src/postprocess.py:7-29: mean over utterances of the fraction of label
    positions (up to the first pad) predicted correctly."""
    predict = np.argmax(predict.detach().cpu().numpy(), axis=-1)
    label = label.cpu().numpy()
    accs = []
    for p, l in zip(predict, label):
        correct, total = 0.0, 0
        for pp, ll in zip(p, l):
            if ll == 0:
                break
            correct += int(pp == ll)
            total += 1
        accs.append(correct / total)
    return sum(accs) / len(accs)


def calc_err(predict, label, mapper):
    """src/postprocess.py:31-49: word-level edit distance over label words."""
    label = label.cpu()
    predict = np.argmax(predict.detach().cpu().numpy(), axis=-1)
    hyp = [mapper.translate(p) for p in predict]
    ref = [mapper.translate(l) for l in label]
    ds = [float(edit_distance(p.split(' '), l.split(' '))) / len(l.split(' '))
          for p, l in zip(hyp, ref)]
    return sum(ds) / len(ds)


def draw_att(att_maps, hyps):
    """src/postprocess.py:51-62: [3, len, T'] image per utterance."""
    out = []
    for i in range(att_maps.shape[0]):
        att_i = att_maps[i, :, :]
        n = len(trim_eos(hyps[i]))
        out.append(torch.stack([att_i, att_i, att_i], dim=0)[:, :n, :])
    return out
