"""Exponential moving average of parameters with the `torch_ema.ExponentialMovingAverage` interface the reference uses
(train_double_latent_semantic.py:145-146, :456-457, :487-488; render_multiview_images_double_semantic.py:62-64 loads a
pickled instance and calls `.copy_to(generator.parameters())`).  torch_ema is not a dependency of this package:
`fenerf_amd.compat.install_aliases()` registers this module under the names `torch_ema` / `torch_ema.ema` when the real
package is absent, so the reference's pickled `*_ema.pth` files (plain objects with `decay`, `num_updates`,
`shadow_params`, `collected_params`) unpickle into this class."""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))   # warm-up of the average
        with torch.no_grad():
            live = [p for p in parameters if p.requires_grad]
            for s, p in zip(self.shadow_params, live):
                s.sub_((1.0 - decay) * (s - p.to(s.device)))

    def copy_to(self, parameters):
        live = [p for p in parameters if p.requires_grad]
        with torch.no_grad():
            for s, p in zip(self.shadow_params, live):
                p.copy_(s)            # (not p.data.copy_: keeps the version counter honest for the native re-pack)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters if p.requires_grad]

    def restore(self, parameters):
        live = [p for p in parameters if p.requires_grad]
        with torch.no_grad():
            for c, p in zip(self.collected_params, live):
                p.copy_(c)

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def load_state_dict(self, state):
        self.decay, self.num_updates = state["decay"], state["num_updates"]
        self.shadow_params, self.collected_params = state["shadow_params"], state["collected_params"]
