"""The two-level evaluation harness of the reference's PNHigh driver
(/root/reference/src/models/trainPNHigh.py): ``SCDataset`` (:15-41) and the eval block of
``TrainModel.train_and_validate`` (:131-144) — Low greedy -> latent -> High greedy per batch of 128,
actions accumulated as ``allActions[T][nTest][8]`` (the artefact ``ML2PN.check`` reads).  The training
loop (:76-112) is gnnpn_sc_amd/trainPNHigh.py.
"""
import torch

from . import ops
from .modelPN import two_level_greedy


class SCDataset(torch.utils.data.Dataset):
    """trainPNHigh.py:15-41.  ``dataset`` = loadDataPN rows [P][L][9]; with embeddingTag=False column 0
    (the category id) is dropped -> FloatTensor [L,8] per problem; with embeddingTag=True the rows stay [L,9] (:26-29) for
    a CombinatorialRL built with embedding_size != 0."""

    def __init__(self, dataset, targets, embeddingTag=False):
        super().__init__()
        self.data_set = [torch.FloatTensor([row if embeddingTag else row[1:] for row in rows]) for rows in dataset]
        self.label = list(targets)
        self.serviceNumbers = [0] * len(self.data_set)
        self.size = len(self.data_set)

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        return self.data_set[idx], self.label[idx]


@torch.no_grad()
def evaluate(low_model, model, val_dataset, serCategory, batch_size=128, device="cuda:0", precision="f32"):
    """trainPNHigh.py:131-144: -> (allActions [T][n][8] python lists, val_tour = mean R per batch).  ``precision``: "f32" by
    default (the artefact / parity path), "split" = the exact three-piece products bench.py measures."""
    dev = torch.device(device)
    all_actions = [[] for _ in range(serCategory)]
    val_tour = []
    for lo in range(0, len(val_dataset), batch_size):
        items = [val_dataset[i][0] for i in range(lo, min(len(val_dataset), lo + batch_size))]
        inputs = torch.stack(items).to(dev)
        def batch_pass(attempt):
            out = two_level_greedy(low_model, model, inputs, write_through=attempt > 0, precision=precision)
            return out["actions"].cpu().numpy(), float(out["R"].mean().item())
        # a timed-out inter-workgroup hand-off must never reach the caller's artefacts: checked per batch, repeated once if it happens
        act, r_mean = ops.run_checked(batch_pass, dev)
        for a in range(serCategory):
            all_actions[a] += act[:, a, :].tolist()
        val_tour.append(r_mean)
    return all_actions, val_tour
