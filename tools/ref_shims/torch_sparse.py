"""Pure-torch stand-in for torch_sparse.SparseTensor (generator-side tooling only)."""
import torch


class SparseTensor:
    def __init__(self, row, col, value, sparse_sizes):
        self.row, self.col, self.value = row, col, value
        self.sizes = tuple(int(s) for s in sparse_sizes)

    def size(self, i):
        return self.sizes[i]

    def coo(self):
        return self.row, self.col, self.value

    def __matmul__(self, x):
        out = x.new_zeros((self.sizes[0],) + tuple(x.shape[1:]))
        return out.index_add(0, self.row, self.value.view(-1, 1) * x[self.col])
