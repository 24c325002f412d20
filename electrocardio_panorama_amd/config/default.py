"""Default configuration tree.  Key names and default values follow reference codes/config/default.py:4-55 (they are
the drop-in surface: `cfg.SOLVER.lr`, `cfg.DATA.lead_num`, ...); the tree is declared once as a nested literal."""
from .cfgnode import CfgNode

_DEFAULTS = {
    "seed": 123,
    "fit_msg": "None",
    "output_dir": "{your folder}",
    "latent_save_dir": "{your folder}",
    "desc": "model_v2_tianchi",
    "DATA": {
        "dataset": "tianchi",
        "train_label_path": "data/tianchi/tianchi_train_jsons.txt",
        "test_label_path": "data/tianchi/tianchi_test_jsons.txt",
        "train_data_root": "data/tianchi/npy_data/tianchi_train_round1",
        "train_label_root": "data/tianchi/tianchi_interval",
        "train_pkl_path": "data/PTB/pkl_data/train_heartbeats.pkl",
        "test_pkl_path": "data/PTB/pkl_data/test_heartbeats.pkl",
        "noise_std": [4.37258895, 4.73799667, 5.00643047, 6.7582663, 6.57354042, 6.31023917, 6.05944371, 7.05612394],
        "lead_num": 1,                 # V: number of input leads (views)
        "noise": False,                # add the per-lead noise sample to the prediction before the loss
        "train_data_mode": "normal",
        "super_mode": "normal",
        "weighted_sample": False,
        "synthetic": False,            # (not in the reference) train / validate on seeded synthetic `meta` batches
    },
    "MODEL": {
        "model": "modelv2",            # 'model_nefnet' selects Nef-Net
        "resume": "",
        "loss": "v1",                  # 'v1' = losswrapper (L1/L2 + Standin terms)
        "jitter_factor": 0.0,          # view-angle jitter in degrees (dataset side)
        "theta_L": 1,
    },
    "SOLVER": {
        "optim": "sgd",
        "scheduler": "steplr",
        "lr_step": [150, 350],
        "lr": 1e-3,
        "epochs": 500,
        "OurLoss1_version": "v2",
        "reg_loss": "l1_loss",         # 'l1_loss' | 'l2_loss' for the reconstruction term
        "loss_using": [1, 2, 3],
        "part_loss_no_grad": False,
        "loss_factor": [1, 1, 1],
        # train step as one captured hipGraph (graph.GraphedTrainStep): None / 'auto' = replay at EVERY batch size wherever the step
        # qualifies (plain Model_nefnet train path, FusedSGD with one parameter group, no DATA.noise, per-view host lists not
        # wanted: Solver._graphed_step; rounds 1-3 replayed launch-bound shapes only); False = always issue eagerly; True = as auto
        "graph": None,
    },
}


def get_defaults():
    """A fresh, mutable copy of the default tree."""
    return CfgNode(_DEFAULTS).clone()


cfg = get_defaults()
