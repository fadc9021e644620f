"""Seeded synthetic inputs for tests and bench.py.

R-MAT (a, b, c, d) = (0.57, 0.19, 0.19, 0.05), no symmetrisation, duplicates removed.
Generation is setup work outside every timed region; it runs on whatever device it
is given (index sort/unique here is ATen plumbing for *input construction*, not part
of the product path).
"""
import torch


def rmat_edges(scale, edge_factor, seed=0, device='cpu', a=0.57, b=0.19, c=0.19,
               row_offset_bits=None):
    """Raw (row, col) draws of an R-MAT graph with 2**scale vertices."""
    n_edges = int(edge_factor * (1 << scale))
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.zeros(n_edges, dtype=torch.int64, device=device)
    col = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_edges, generator=g, device=device)
        rbit = (u >= a + b).to(torch.int64)
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        row = (row << 1) | rbit
        col = (col << 1) | cbit
    return row, col


def uniform_edges(m, n, n_edges, seed=0, device='cpu'):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    row = torch.randint(0, m, (n_edges, ), generator=g, device=device)
    col = torch.randint(0, n, (n_edges, ), generator=g, device=device)
    return row, col


def to_csr(row, col, m, n):
    """Sort row-major, drop duplicates -> (rowptr, col) int64.  Setup only."""
    key = torch.unique(row * n + col)  # sorted + deduplicated
    row = torch.div(key, n, rounding_mode='floor')
    col = key - row * n
    counts = torch.bincount(row, minlength=m)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=row.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, col


def rmat_csr(scale, edge_factor, seed=0, device='cpu'):
    n = 1 << scale
    row, col = rmat_edges(scale, edge_factor, seed, device)
    return to_csr(row, col, n, n)


def uniform_degree_csr(m, n, deg, seed=0, device='cpu'):
    """Control graph: exactly `deg` entries per row, uniform columns (may repeat)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    col = torch.randint(0, n, (m, deg), generator=g, device=device)
    col, _ = torch.sort(col, dim=1)
    rowptr = torch.arange(0, (m + 1) * deg, deg, dtype=torch.int64, device=device)
    return rowptr, col.reshape(-1).contiguous()


def values(n, seed=1, dtype=torch.float32, device='cpu'):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand(n, generator=g, device=device).to(dtype)


def features(n, f, seed=2, dtype=torch.float32, device='cpu', batch=()):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randn(*batch, n, f, generator=g, device=device).to(dtype)
