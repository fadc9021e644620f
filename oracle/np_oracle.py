"""numpy restatement of the reference's Python-level pipelines -- TEST INFRASTRUCTURE ONLY.

  sort_coo / csr2csc      torch_sparse/storage.py:149-162, 407-416 (key = row * N + col, sort)
  coalesce                torch_sparse/storage.py:436-466 + torch_scatter.segment_csr semantics
  transpose               torch_sparse/transpose.py:39-62
  spspmm                  torch_sparse/matmul.py:94-111 -> torch.sparse.mm; PyTorch's SpGEMM is not
                          under /root/reference (third-party: torch, unpinned by the reference,
                          2.10.0 here), so this restates the published row-wise (Gustavson)
                          definition: C[i, :] = sum_k A[i, k] * B[k, :], rows sorted, duplicates summed.

Pinned by tests/test_oracle.py against the fixtures written by tests/golden/make_golden.py (which
ran the reference's own Python) -- see that file.  The reference's sort is not stable; this one
is, which only matters for the order in which duplicate values are reduced.
"""
import numpy as np


def sort_coo(row, col, m, n):
    key = np.asarray(row, dtype=np.int64) * n + np.asarray(col, dtype=np.int64)
    perm = np.argsort(key, kind='stable')
    return np.asarray(row)[perm], np.asarray(col)[perm], perm


def csr2csc(row, col, m, n):
    key = np.asarray(col, dtype=np.int64) * m + np.asarray(row, dtype=np.int64)
    return np.argsort(key, kind='stable')


def coalesce(row, col, value, m, n, op='add'):
    row, col, perm = sort_coo(row, col, m, n)
    key = row.astype(np.int64) * n + col
    head = np.ones(key.shape, dtype=bool)
    head[1:] = key[1:] != key[:-1]
    starts = np.nonzero(head)[0]
    out_row, out_col = row[head], col[head]
    if value is None:
        return out_row, out_col, None
    v = np.asarray(value)[perm]
    if starts.size == 0:
        return out_row, out_col, v[:0]
    if op in ('add', 'sum'):
        out = np.add.reduceat(v, starts, axis=0)
    elif op == 'mean':
        cnt = np.diff(np.append(starts, key.size))
        s = np.add.reduceat(v, starts, axis=0)
        cnt = cnt.reshape((-1, ) + (1, ) * (v.ndim - 1))
        out = np.floor_divide(s, cnt) if np.issubdtype(v.dtype, np.integer) else s / cnt
    elif op == 'min':
        out = np.minimum.reduceat(v, starts, axis=0)
    elif op == 'max':
        out = np.maximum.reduceat(v, starts, axis=0)
    else:
        raise ValueError(op)
    return out_row, out_col, out.astype(v.dtype)


def transpose(row, col, value, m, n):
    return coalesce(col, row, value, n, m, 'add')


def spspmm(rowA, colA, valA, rowB, colB, valB, m, k, n):
    """Row-wise SpGEMM on coalesced COO inputs; returns sorted (row, col, val)."""
    rowA, colA, rowB, colB = (np.asarray(x, dtype=np.int64) for x in (rowA, colA, rowB, colB))
    valA = np.ones(rowA.size) if valA is None else np.asarray(valA)
    valB = np.ones(rowB.size) if valB is None else np.asarray(valB)
    ptrB = np.zeros(k + 1, dtype=np.int64)
    np.add.at(ptrB, rowB + 1, 1)
    ptrB = np.cumsum(ptrB)
    cnt = ptrB[colA + 1] - ptrB[colA]
    total = int(cnt.sum())
    ea = np.repeat(np.arange(rowA.size), cnt)
    off = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    eb = ptrB[colA[ea]] + off
    prow, pcol = rowA[ea], colB[eb]
    pval = (valA[ea] * valB[eb]).astype(np.result_type(valA.dtype, valB.dtype))
    return coalesce(prow, pcol, pval, m, n, 'add')


# ---------------------------------------------------------------------------------------------
# mini-batch producers -- pinned by tests/golden/py4_*.npz, the outputs of
# the reference's csrc/cpu/{rw,sample,saint,relabel}_cpu.cpp compiled unmodified
# ---------------------------------------------------------------------------------------------
def random_walk(rowptr, col, start, rand):
    """csrc/cpu/rw_cpu.cpp:29-42: cur = col[row_start + int64(rand * deg)] with a float32 product.
    (Nodes without neighbours keep the walk in place; the reference reads out of the row there.)"""
    rowptr, col = np.asarray(rowptr, np.int64), np.asarray(col, np.int64)
    rand = np.asarray(rand, np.float32)
    n, L = rand.shape
    out = np.empty((n, L + 1), np.int64)
    cur = np.asarray(start, np.int64).copy()
    out[:, 0] = cur
    for l in range(L):
        s = rowptr[cur]
        deg = rowptr[cur + 1] - s
        p = (rand[:, l] * deg.astype(np.float32)).astype(np.int64)
        p = np.minimum(p, np.maximum(deg - 1, 0))
        has = deg > 0
        cur = np.where(has, col[np.where(has, s + p, 0)] if col.size else cur, cur)
        out[:, l + 1] = cur
    return out


def relabel(col, idx):
    """csrc/cpu/relabel_cpu.cpp:5-46: ids in idx keep their position, every other id gets
    len(idx) + rank of its first occurrence in col."""
    col, idx = np.asarray(col, np.int64), np.asarray(idx, np.int64)
    M = int(max(col.max(initial=-1), idx.max(initial=-1))) + 1
    local = np.full(M, -1, np.int64)
    local[idx] = np.arange(idx.size)
    is_new = local[col] < 0 if col.size else np.zeros(0, bool)
    uniq, first = np.unique(col[is_new], return_index=True)
    order = np.argsort(first, kind='stable')
    new_nodes = uniq[order]
    local[new_nodes] = idx.size + np.arange(new_nodes.size)
    return (local[col] if col.size else col), np.concatenate([idx, new_nodes])


def _gather_rows(rowptr, idx):
    rowptr, idx = np.asarray(rowptr, np.int64), np.asarray(idx, np.int64)
    cnt = rowptr[idx + 1] - rowptr[idx] if idx.size else np.zeros(0, np.int64)
    out_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    seg = np.repeat(np.arange(idx.size), cnt)
    pos = np.arange(out_ptr[-1]) - out_ptr[seg] + rowptr[idx][seg] if idx.size else np.zeros(0, np.int64)
    return out_ptr, seg, pos.astype(np.int64)


def relabel_one_hop(rowptr, col, idx, bipartite):
    """csrc/cpu/relabel_cpu.cpp:48-155 -> (out_rowptr, out_col, positions of the kept entries, out_idx)"""
    out_ptr, _, pos = _gather_rows(rowptr, idx)
    out_col, out_idx = relabel(np.asarray(col, np.int64)[pos], idx)
    if not bipartite:
        extra = out_idx.size - np.asarray(idx).size
        out_ptr = np.concatenate([out_ptr, np.full(extra, pos.size, np.int64)])
    return out_ptr, out_col, pos, out_idx


def sample_adj_all(rowptr, col, idx):
    """csrc/cpu/sample_cpu.cpp:40-58,116-137 with num_neighbors < 0: every neighbour, relabelled in
    first-occurrence order, rows sorted by the new column id -> (rowptr, col, n_id, e_id)."""
    out_ptr, seg, pos = _gather_rows(rowptr, idx)
    local, n_id = relabel(np.asarray(col, np.int64)[pos], idx)
    order = np.lexsort((local, seg))
    return out_ptr, local[order], n_id, pos[order]


def saint_subgraph(idx, rowptr, col):
    """csrc/cpu/saint_cpu.cpp:5-52 -> (row, col, edge_index)"""
    idx = np.asarray(idx, np.int64)
    assoc = np.full(np.asarray(rowptr).size - 1, -1, np.int64)
    assoc[idx] = np.arange(idx.size)
    _, seg, pos = _gather_rows(rowptr, idx)
    w = assoc[np.asarray(col, np.int64)[pos]] if pos.size else np.zeros(0, np.int64)
    keep = w >= 0
    return seg[keep], w[keep], pos[keep]


def neighbor_sample_all(colptr, row, input_node, num_hops, directed):
    """csrc/cpu/neighbor_sample_cpu.cpp:15-124 with num_neighbors = [-1] * num_hops (every in-neighbour):
    -> (node, row, col, edge).  Hop l expands the nodes found in hop l-1; new nodes are appended in
    first-occurrence order; directed=False returns every stored edge between the sampled nodes."""
    colptr, row = np.asarray(colptr, np.int64), np.asarray(row, np.int64)
    samples = np.asarray(input_node, np.int64)
    begin, end = 0, samples.size
    rows, cols, edges = [], [], []
    for _ in range(num_hops):
        frontier = samples[begin:end]
        _, seg, pos = _gather_rows(colptr, frontier)
        local, samples = relabel(row[pos], samples)
        rows.append(local), cols.append(seg + begin), edges.append(pos)
        begin, end = end, samples.size
    if not directed:
        i, v, pos = saint_subgraph(samples, colptr, row)
        return samples, v, i, pos
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)  # noqa: E731
    return samples, cat(rows), cat(cols), cat(edges)


def hetero_neighbor_sample_det(node_types, edge_types, colptr, row, inputs, fan, num_hops, directed, node_time=None):
    """csrc/cpu/neighbor_sample_cpu.cpp:135-430 (hetero_sample) for the draws that are NOT random: fan < 0, or at
    least as many draws as the node has neighbours ("select all neighbors", :253-293) -- any other node raises.
    colptr / row: {relation: array} with relation = 'src__rel__dst'; inputs: {node type: ids}; fan: {relation: [k per
    hop]}; node_time: {node type: times} makes it the temporal sampler (:169-200, 263-280: nodes are (node, root)
    pairs, a neighbour v counts only if time[v] <= root time of the node it is drawn for; types without times are
    unconstrained, :119-130).  Sequential dict model, statement by statement: relations in sorted key order per hop
    (:216-219), frontier slices moved at the end of a hop (:352-361), map INSERT semantics (a seed listed twice keeps
    its first position, :192-195), the undirected edge list at the end (:366-412).
    -> (node {type: ids}, row, col, edge {relation: ids})."""
    temporal = node_time is not None
    to_edge = {'__'.join(e): e for e in edge_types}
    samples = {t: [] for t in node_types}          # ids, or (id, root) pairs when temporal
    to_local = {t: {} for t in node_types}
    root_time = {t: [] for t in node_types}
    rows = {r: [] for r in colptr}
    cols = {r: [] for r in colptr}
    edges = {r: [] for r in colptr}
    for t, x in inputs.items():
        for i, v in enumerate(np.asarray(x, np.int64).tolist()):
            key = (v, i) if temporal else v
            samples[t].append(key)
            to_local[t].setdefault(key, i)
            if temporal:
                root_time[t].append(int(node_time[t][v]))
    slices = {t: (0, len(samples[t])) for t in node_types}
    for ell in range(num_hops):
        for rel in sorted(fan):
            src_t, _, dst_t = to_edge[rel]
            k = fan[rel][ell]
            cp, rw = np.asarray(colptr[rel], np.int64), np.asarray(row[rel], np.int64)
            begin, end = slices[dst_t]
            for i in range(begin, end):
                w = samples[dst_t][i][0] if temporal else samples[dst_t][i]
                root_w = samples[dst_t][i][1] if temporal else -1
                dst_time = root_time[dst_t][i] if temporal else 0
                s, e = int(cp[w]), int(cp[w + 1])
                if e == s:
                    continue
                if not (k < 0 or k >= e - s):
                    raise ValueError('random draw: node %d of %s has %d neighbours, fan-out %d' % (w, dst_t, e - s, k))
                for off in range(s, e):
                    v = int(rw[off])
                    if temporal:
                        if src_t in node_time and not int(node_time[src_t][v]) <= dst_time:
                            continue
                        key = (v, root_w)
                        if key not in to_local[src_t]:
                            to_local[src_t][key] = len(samples[src_t])
                            samples[src_t].append(key)
                            root_time[src_t].append(dst_time)
                        cols[rel].append(i), rows[rel].append(to_local[src_t][key]), edges[rel].append(off)
                    else:
                        if v not in to_local[src_t]:
                            to_local[src_t][v] = len(samples[src_t])
                            samples[src_t].append(v)
                        if directed:
                            cols[rel].append(i), rows[rel].append(to_local[src_t][v]), edges[rel].append(off)
        slices = {t: (slices[t][1], len(samples[t])) for t in node_types}
    if not directed:
        assert not temporal
        for rel in colptr:
            src_t, _, dst_t = to_edge[rel]
            cp, rw = np.asarray(colptr[rel], np.int64), np.asarray(row[rel], np.int64)
            for i, w in enumerate(samples[dst_t]):
                for off in range(int(cp[w]), int(cp[w + 1])):
                    j = to_local[src_t].get(int(rw[off]))
                    if j is not None:
                        rows[rel].append(j), cols[rel].append(i), edges[rel].append(off)
    arr = lambda x: np.asarray(x, np.int64)  # noqa: E731
    node = {t: arr([p[0] for p in samples[t]] if temporal else samples[t]) for t in node_types}
    return node, {r: arr(v) for r, v in rows.items()}, {r: arr(v) for r, v in cols.items()}, {r: arr(v) for r, v in edges.items()}

