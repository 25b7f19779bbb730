"""The reference's MLP benchmark kernel expressed as xsmm dispatch/invoke calls, and
its single-node multi-GPU sharding.

What the reference does (tools/mlir-gen/MLIRGen.cpp:632-681, lowered by the default
pipeline): every layer `relu(X @ W + bias)` becomes independent output tiles, each
one xsmm.fused_brgemm [add(bcast_col_in0), relu] (CombineXsmmPass.cpp:31-145), run
under scf.parallel / OpenMP over the tile grid (DefaultPipeline.cpp:179-180).

Here a layer is ONE whole-layer fused_brgemm dispatch per rank (the HIP kernel tiles
it over the CUs itself), and the tile grid is sharded across GPUs by ROW BLOCKS of
the batch dimension: rank r owns rows [row0, row0 + rows) of the activations for ALL
layers, weights and biases are replicated, there is no inter-layer traffic, and one
all-gather of the final activations (RCCL over xGMI) rebuilds the full output.
The reference has no distributed path at all: this module is
the new component, the per-tile arithmetic is unchanged.
"""
from dataclasses import dataclass, field
from typing import List

from .runtime import BinaryFlags, BinaryKind, DataType, GemmFlags, UnaryKind

K_CHUNK = 64  # k per batch-reduce step of a whole-layer dispatch


@dataclass
class MlpSpec:
    batch: int = 4096
    layers: List[int] = field(default_factory=lambda: [1024, 1024, 1024, 1024])
    dtype: int = DataType.BF16
    bias: bool = True
    relu: bool = True

    @property
    def vnni(self):
        return self.dtype == DataType.BF16

    def flops(self):
        """BENCH_TOTAL_FLOPS of mlir-gen (MLIRGen.cpp:313-334): 2MNK (+MN bias) (+MN relu) per layer"""
        total = 0
        for k, n in zip(self.layers[:-1], self.layers[1:]):
            total += 2 * self.batch * n * k
            total += self.batch * n if self.bias else 0
            total += self.batch * n if self.relu else 0
        return total


def row_partition(batch, world, rank, granule=128):
    """contiguous row block of `rank`: blocks are multiples of `granule` rows (whole
    output tiles stay on one GPU); earlier ranks take the remainder blocks"""
    if batch % granule:
        granule = 1
    blocks = batch // granule
    base, extra = divmod(blocks, world)
    mine = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first * granule, mine * granule


def layer_dispatch_args(spec, rows, k, n):
    """argument tuple of xsmm_fused_brgemm_dispatch for one whole layer on `rows` rows"""
    assert k % K_CHUNK == 0
    gemm_flags = GemmFlags.BETA_0 | (GemmFlags.VNNI_B if spec.vnni else 0)
    return dict(dtype=spec.dtype, m=rows, n=n, k=K_CHUNK, lda=k, ldb=n, ldc=n, stride_a=K_CHUNK,
                stride_b=K_CHUNK * n, gemm_flags=gemm_flags, unary_flags=0,
                unary_kind=UnaryKind.RELU if spec.relu else UnaryKind.NONE,
                binary_flags=BinaryFlags.BCAST_COL_IN_0 if spec.bias else BinaryFlags.NONE,
                binary_kind=BinaryKind.ADD if spec.bias else BinaryKind.NONE), k // K_CHUNK


class ShardedMlp:
    """One rank's share of the MLP. `rt` is an XsmmRuntime (product) - tests inject an
    object with the same fused_brgemm_dispatch / fused_brgemm methods."""

    def __init__(self, spec, rank=0, world=1, rt=None, chain=True):
        self.spec, self.rank, self.world, self.rt = spec, rank, world, rt
        self.chain, self.last_step_fused = chain, False
        self.row0, self.rows = row_partition(spec.batch, world, rank)
        self.handles = []
        for k, n in zip(spec.layers[:-1], spec.layers[1:]):
            args, br = layer_dispatch_args(spec, self.rows, k, n)
            h = rt.fused_brgemm_dispatch(**args) if self.rows > 0 else 0
            self.handles.append((h, br))

    def forward(self, x_local, weights, biases, acts):
        """x_local: this rank's rows of the input [rows, layers[0]]; weights[l]: [K][N]
        (f32) or VNNI-2 [K/2][N][2] (bf16); biases[l]: [N]; acts[l]: output buffer of
        layer l [rows, layers[l+1]]. Only enqueues work (async mode) or runs it (sync)."""
        if self.rows == 0:
            return None
        cur = x_local
        dummy = biases[0] if biases else cur
        calls = []
        for l, (h, br) in enumerate(self.handles):
            d = biases[l] if self.spec.bias else dummy
            calls.append((h, cur, 0, weights[l], 0, acts[l], 0, d, 0, br))
            cur = acts[l]
        if self.chain and hasattr(self.rt, "fused_brgemm_chain"):
            # ONE call for the rank's whole step: the runtime runs the layer chain as a single persistent launch when it can
            # (bf16, device buffers, async mode: xsmm_hip_fused_brgemm_chain_invoke), else call by call - same result
            self.last_step_fused = self.rt.fused_brgemm_chain(self.spec.dtype, calls)
        else:
            self.last_step_fused = False
            for c in calls:
                self.rt.fused_brgemm(self.spec.dtype, *c)
        return cur


class ColumnShardedMlp:
    """The other single-node sharding of the same tile grid ("all-gather of
    the output of a layer"): rank r computes the COLUMN block [n0, n0 + N/W) of every layer for the
    whole batch - it reads only its 1/W of each weight matrix - and the activations are all-gathered
    after EACH layer. The gathered activations keep the collective's natural layout
    G[W][batch][N/W] (rank-major, exactly what all_gather_into_tensor produces, no re-layout
    kernel): the next layer consumes it directly, because the rank dimension is a batch-reduce
    dimension - ONE brgemm dispatch with k = N/W per batch element, lda = N/W,
    stride_a = batch * N/W, stride_b = (N/W) * N_next, br = W sums over the W column blocks.
    Per step: L collectives of batch * N / W elements per rank, against one for the row sharding."""

    def __init__(self, spec, rank=0, world=1, rt=None):
        self.spec, self.rank, self.world, self.rt = spec, rank, world, rt
        for n in spec.layers[1:]:
            if n % world or (spec.vnni and (n // world) % 2):
                raise ValueError("layer width %d cannot be split into %d column blocks" % (n, world))
        self.handles = []
        gemm_flags = GemmFlags.BETA_0 | (GemmFlags.VNNI_B if spec.vnni else 0)
        for l, (k, n) in enumerate(zip(spec.layers[:-1], spec.layers[1:])):
            nw = n // world
            if l == 0:  # the input is row-major [batch][k] on every rank: k chunks of 64 as usual
                assert k % K_CHUNK == 0
                kk, lda, sa, sb, br = K_CHUNK, k, K_CHUNK, K_CHUNK * n, k // K_CHUNK
            else:       # gathered layout [W][batch][k/W]: one batch element per rank block
                kk, lda, sa, sb, br = k // world, k // world, spec.batch * (k // world), (k // world) * n, world
            h = rt.fused_brgemm_dispatch(
                dtype=spec.dtype, m=spec.batch, n=nw, k=kk, lda=lda, ldb=n, ldc=nw, stride_a=sa, stride_b=sb,
                gemm_flags=gemm_flags, unary_flags=0, unary_kind=UnaryKind.RELU if spec.relu else UnaryKind.NONE,
                binary_flags=BinaryFlags.BCAST_COL_IN_0 if spec.bias else BinaryFlags.NONE,
                binary_kind=BinaryKind.ADD if spec.bias else BinaryKind.NONE)
            self.handles.append((h, br, nw))

    def forward(self, x, weights, biases, locals_, gathered, all_gather):
        """x: [batch, layers[0]] on every rank; weights[l]: the FULL [K][N] (f32) or VNNI-2 [K/2][N][2]
        (bf16) matrix (only this rank's columns are read); biases[l]: [N]; locals_[l]: this rank's
        output block [batch, N/W]; gathered[l]: [W, batch, N/W]; all_gather(dst, src) performs the
        collective (dist.all_gather_into_tensor on RCCL / gloo). Returns gathered[-1]."""
        cur = x
        for l, (h, br, nw) in enumerate(self.handles):
            n0 = self.rank * nw
            d = biases[l] if self.spec.bias else biases[0]
            self.rt.fused_brgemm(self.spec.dtype, h, cur, 0, weights[l], (2 if self.spec.vnni else 1) * n0,
                                 locals_[l], 0, d, n0, br)
            all_gather(gathered[l], locals_[l])
            cur = gathered[l]
        return cur


def gathered_to_rows(g):
    """[W, batch, N/W] (rank-major column blocks) -> row-major [batch, N]; for consumers outside the MLP"""
    w, b, nw = g.shape
    return g.permute(1, 0, 2).reshape(b, w * nw)


def all_gather_rows(out_local, out_full, spec, world, group=None):
    """all-gather of the row blocks into the full [batch, N] output. Equal blocks use
    one all_gather_into_tensor (a single RCCL collective on GPUs); ragged blocks fall
    back to all_gather on a padded list."""
    import torch.distributed as dist

    sizes = [row_partition(spec.batch, world, r)[1] for r in range(world)]
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(out_full, out_local, group=group)
        return out_full
    n = out_full.shape[1]
    mx = max(sizes)
    import torch

    pad = torch.zeros((mx, n), dtype=out_local.dtype, device=out_local.device)
    pad[: out_local.shape[0]] = out_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    row = 0
    for r in range(world):
        out_full[row: row + sizes[r]] = parts[r][: sizes[r]]
        row += sizes[r]
    return out_full
