class AverageMeter:
    """running means keyed by name (reference training/utils.py:1-25)"""

    def __init__(self) -> None:
        self.keys: list[str] = []
        self.avg: dict[str, float] = {}
        self.sum: dict[str, float] = {}
        self.count: dict[str, int] = {}

    def reset(self) -> None:
        for k in self.keys:
            self.avg[k], self.sum[k], self.count[k] = 0, 0, 0

    def update(self, val: float, key: str, n: int = 1) -> None:
        if key not in self.keys:
            self.keys.append(key)
            self.sum[key], self.count[key] = 0.0, 0
        self.sum[key] += val * n
        self.count[key] += n
        self.avg[key] = self.sum[key] / self.count[key]
