import torch
import torch.nn.functional as F


def gelu(x):
    return F.gelu(x.float()).type_as(x)


def get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return gelu
    if activation == "tanh":
        return torch.tanh
    if activation == "linear":
        return lambda x: x
    if activation == "swish":
        return F.silu
    raise RuntimeError("--activation-fn {} not supported".format(activation))


def index_put(tensor, indices, value):
    tensor[indices] = value
    return tensor
