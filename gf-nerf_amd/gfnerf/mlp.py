"""MLPNetwork (gfnerf/mlp.py:3-57): nn.Linear stack with bias, ReLU hidden, optional Sigmoid output; the layers run
on the f32-MFMA kernels (gfn_linear_fwd / gfn_linear_bwd) through one autograd Function per layer."""
import torch
from torch import nn

from .. import ops


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        x = x.contiguous()
        y = ops.linear_fwd(x, W.contiguous(), b.contiguous(), act)
        ctx.save_for_backward(x, W, y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dx, dW, db = ops.linear_bwd(x, W.contiguous(), y, dy.contiguous(), ctx.act, need_dx=ctx.needs_input_grad[0])
        return dx, dW, db, None


class MLPNetwork(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = network_config
        self.n_neurons = network_config["n_neurons"]
        self.n_hidden_layers = network_config["n_hidden_layers"]
        self.sigmoid_out = network_config["output_activation"] == "Sigmoid"
        assert network_config["activation"] == "ReLU"
        self.layers = nn.ModuleList()
        for i in range(self.n_hidden_layers + 1):
            n_in = self.n_input_dims if i == 0 else self.n_neurons
            n_out = self.n_output_dims if i == self.n_hidden_layers else self.n_neurons
            self.layers.append(nn.Linear(n_in, n_out))

    def forward(self, x):
        shape = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1]).float()
        for i, layer in enumerate(self.layers):
            last = i == self.n_hidden_layers
            act = ("sigmoid" if self.sigmoid_out else "none") if last else "relu"
            x = _LinearFn.apply(x, layer.weight, layer.bias, act)
        return x.reshape(*shape, -1)
