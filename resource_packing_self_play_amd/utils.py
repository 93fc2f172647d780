"""Small helpers with the reference's names (xw_mcts/utils.py:1-22)."""


class AverageMeter:
    """Running mean of a scalar; prints as the mean with two significant decimals."""

    def __init__(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __repr__(self):
        return "%.2e" % self.avg


class dotdict(dict):
    """dict with attribute access, the `args` object the reference passes around (main_bpp.py:54)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value
