"""Loss primitives (oracle; test infrastructure only).

recbole 1.0.1 (un-vendored):
  BPRLoss(gamma=1e-10): -log(gamma + sigmoid(pos - neg)).mean()
  EmbLoss(norm=2)(*embs, require_pow=False): sum_e ||e||_F / embs[-1].shape[0]   -> shape [1]
Call sites: emcdr.py:54,80,118-120,126-130 ; cmf.py:47-48,93-98 ; bitgcf.py:69,233,245.
Stock torch: nn.MSELoss (emcdr.py:50,81), nn.BCELoss (cmf.py:45, conet.py:63, bitgcf.py:67),
nn.TripletMarginLoss(margin) (sscdr.py:69).
"""
import torch
import torch.nn.functional as F

BPR_GAMMA = 1e-10


def bpr_loss(pos_score, neg_score, gamma=BPR_GAMMA):
    return -torch.log(gamma + torch.sigmoid(pos_score - neg_score)).mean()


def emb_loss(*embeddings):
    loss = torch.zeros(1)
    for e in embeddings:
        loss = loss + torch.norm(e, p=2)
    return loss / embeddings[-1].shape[0]


def mse_loss(pred, target):
    return F.mse_loss(pred, target)          # mean over ALL elements


def bce_loss(prob, label):
    return F.binary_cross_entropy(prob, label)   # log clamped at -100, mean


def triplet_margin_loss(anchor, positive, negative, margin):
    # defaults p=2, eps=1e-6, swap=False, reduction='mean' (pairwise_distance adds eps to the difference)
    return F.triplet_margin_loss(anchor, positive, negative, margin=margin)
