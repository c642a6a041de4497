"""Abstract inverter interface (reference null_inverter.py:5-15)."""


class NullInverter:
    def __init__(self, model):
        self.model = model

    def to(self, device):
        self.model.to(device)
        return self

    def invert(self, target_img, depth, prompt, num_inner_steps=10, early_stop_epsilon=1e-5, verbose=False):
        raise NotImplementedError("Null inverter must implement invert method.")
