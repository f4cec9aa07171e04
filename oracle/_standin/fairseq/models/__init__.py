import torch.nn as nn


class BaseFairseqModel(nn.Module):
    def __init__(self):
        super().__init__()
