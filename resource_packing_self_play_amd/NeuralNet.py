"""Plugin base class with the reference's method list (xw_mcts/NeuralNet.py:1-50).  The reference's subclasses
changed the signatures (no player argument); these are the ones the bin-packing path really uses."""


class NeuralNet:
    def __init__(self, game, args):
        pass

    def train(self, examples):
        """examples: list of (state, pi, v) (NNet.py:27-30)"""
        raise NotImplementedError

    def predict(self, board):
        """board: (N+1, H, W) state -> (pi float32[A], v float32[1]) (NNet.py:69-85)"""
        raise NotImplementedError

    def save_checkpoint(self, folder, filename):
        raise NotImplementedError

    def load_checkpoint(self, folder, filename):
        raise NotImplementedError
