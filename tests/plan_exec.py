"""numpy executor of the multifrontal plan (TEST-ONLY checker of the host-side structure phase).

It replays, in plain numpy on the CPU, exactly what the HIP kernels k_factor_level /
k_backsolve_level do with the plan that csrc/gs_plan.cpp builds: assemble original H blocks into
dense fronts, extend-add the children's update matrices, partial Cholesky with the rhs carried as an
extra row, backward solve from the root — and, for pose-window shards, the three-phase variant
(own subtrees + contribution to the shared fronts, all-reduce of the exchange buffer, shared top).
Used by the `not gpu` tests to prove the plan (ordering, symbolic factorisation, assembly records,
child maps, shard ownership and exchange layout) exact against the oracle's joint solve without a GPU.
It is not product code and nothing in the package imports it.
"""
import numpy as np


class Plan:
    def __init__(self, flat):
        f = np.asarray(flat, dtype=np.int64)
        assert f[0] == 0x47535031, "bad magic"
        (self.n_scalar, self.n_fronts, self.n_levels, self.max_front, self.n_poses, self.n_lms, self.n_pl,
         self.n_pp, n_asm, n_bnd, n_map, n_child, self.ell_len, self.ell_T, self.ell_R) = [int(v) for v in f[1:16]]
        o = 16

        def take(n, width=1):
            nonlocal o
            a = f[o:o + n * width]
            o += n * width
            return a.reshape(n, width) if width > 1 else a

        self.pose_gidx = take(self.n_poses)
        self.lm_gidx = take(self.n_lms)
        self.pl_order = take(self.n_pl)
        self.pp_order = take(self.n_pp)
        self.ell_ins = take(self.ell_len)       # device (ELL) index of an observation edge -> insertion index, -1 empty
        fr = take(self.n_fronts, 13)
        (self.npiv, self.nbnd, self.piv0, self.parent, self.level, self.owner, self.bnd_off, self.map_off,
         self.asm_off, self.asm_cnt, self.asm_dup, self.child_off, self.child_cnt) = [fr[:, k] for k in range(13)]
        self.bnd_rows = take(n_bnd)
        self.child_map = take(n_map)
        self.children = take(n_child)
        self.asm = take(n_asm, 4)
        self.level_start = take(self.n_levels + 1)
        self.level_fronts = take(self.n_fronts)
        self.world, self.rank, self.n_shared, self.exchange_doubles = [int(v) for v in take(4)]
        if self.world > 1:
            self.pl_rank = take(self.n_pl); self.pp_rank = take(self.n_pp)
            self.pose_known = take(self.n_poses).astype(bool); self.lm_known = take(self.n_lms).astype(bool)
            self.x_off = take(self.n_fronts)
        assert o == len(f), "trailing data in plan dump"

    # ---- structural invariants every valid plan satisfies
    def check_invariants(self):
        S = self.n_fronts
        assert np.all(self.npiv > 0)
        # pivots tile [0, n_scalar) in order
        assert self.piv0[0] == 0
        assert np.all(self.piv0[1:] == self.piv0[:-1] + self.npiv[:-1])
        assert self.piv0[-1] + self.npiv[-1] == self.n_scalar
        gid = np.concatenate([self.pose_gidx[self.pose_gidx >= 0], self.lm_gidx[self.lm_gidx >= 0]])
        assert len(np.unique(gid)) == len(gid)
        for s in range(S):
            b = self.bnd_rows[self.bnd_off[s]:self.bnd_off[s] + self.nbnd[s]]
            assert np.all(np.diff(b) > 0), "boundary rows not ascending"
            if len(b):
                assert b[0] >= self.piv0[s] + self.npiv[s], "boundary row precedes the pivots"
                p = self.parent[s]
                assert p > s, "parent must come later in the elimination order"
                assert self.piv0[p] <= b[0] < self.piv0[p] + self.npiv[p], "parent does not own the first boundary row"
                assert self.level[p] > self.level[s]
                prow = np.concatenate([np.arange(self.piv0[p], self.piv0[p] + self.npiv[p]),
                                       self.bnd_rows[self.bnd_off[p]:self.bnd_off[p] + self.nbnd[p]]])
                m = self.child_map[self.map_off[s]:self.map_off[s] + self.nbnd[s]]
                assert np.array_equal(prow[m], b), "child map does not land on the same unknowns"
                assert np.all(np.diff(m) > 0)
            else:
                assert self.parent[s] == -1
        # levels partition the fronts
        assert sorted(self.level_fronts.tolist()) == list(range(S))
        for l in range(self.n_levels):
            for s in self.level_fronts[self.level_start[l]:self.level_start[l + 1]]:
                assert self.level[s] == l
        if self.world > 1:
            for s in range(S):
                p = self.parent[s]
                if self.owner[s] < 0:
                    assert p < 0 or self.owner[p] < 0, "a shared front must sit under shared fronts only"
                    assert self.x_off[s] >= 0
                else:
                    assert 0 <= self.owner[s] < self.world and self.x_off[s] < 0
                    assert p < 0 or self.owner[p] in (-1, self.owner[s]), "subtrees of different ranks must not nest"

    # ---- pieces of the numeric replay
    def _device_blocks(self, blocks):
        Hpl = np.zeros((self.ell_len, 6))       # assembly records address observation edges by their ELL index
        if self.n_pl:
            live = self.ell_ins >= 0
            Hpl[live] = blocks["Hpl"][self.ell_ins[live]]
        Hpp_off = blocks["Hpp_off"][self.pp_order] if self.n_pp else blocks["Hpp_off"]
        return Hpl, Hpp_off

    def _assemble(self, s, blocks, Hpl, Hpp_off):
        npv, nb = int(self.npiv[s]), int(self.nbnd[s])
        f = npv + nb
        F = np.zeros((f + 1, f))                # lower triangle + rhs row
        for kind, src, r0, c0 in self.asm[self.asm_off[s]:self.asm_off[s] + self.asm_cnt[s]]:
            if kind == 0:
                H = blocks["Hpp_diag"][src].reshape(3, 3)
                for c in range(3):
                    F[r0 + c:r0 + 3, c0 + c] += H[c:, c]
                F[f, c0:c0 + 3] += blocks["b_pose"][src]
            elif kind in (1, 6):                # 6: a landmark appended after the plan was built (grow_plan), same block
                H = blocks["Hll_diag"][src].reshape(2, 2)
                for c in range(2):
                    F[r0 + c:r0 + 2, c0 + c] += H[c:, c]
                F[f, c0:c0 + 2] += blocks["b_lm"][src]
            elif kind == 2:
                F[r0:r0 + 3, c0:c0 + 3] += Hpp_off[src].reshape(3, 3)
            elif kind == 3:
                F[r0:r0 + 3, c0:c0 + 3] += Hpp_off[src].reshape(3, 3).T
            elif kind == 4:
                F[r0:r0 + 3, c0:c0 + 2] += Hpl[src].reshape(3, 2)
            else:
                F[r0:r0 + 2, c0:c0 + 3] += Hpl[src].reshape(3, 2).T
        return F

    def _extend_add(self, s, F, Us, pick):
        f = int(self.npiv[s] + self.nbnd[s])
        for c in self.children[self.child_off[s]:self.child_off[s] + self.child_cnt[s]]:
            if not pick(int(c)):
                continue
            nbc = int(self.nbnd[c])
            m = self.child_map[self.map_off[c]:self.map_off[c] + nbc]
            rows = np.concatenate([m, [f]])
            U = Us[c]
            for col in range(nbc):
                F[rows[col:], m[col]] += U[col:, col]

    def _factor(self, s, F):
        npv, nb = int(self.npiv[s]), int(self.nbnd[s])
        f = npv + nb
        ok = True
        for k in range(npv):
            piv = F[k, k]
            if not piv > 0:
                ok = False
                piv = 1.0
            d = np.sqrt(piv)
            F[k + 1:, k] /= d
            F[k, k] = d
            for c in range(k + 1, f):
                F[c:, c] -= F[c:, k] * F[c, k]
        return F[:, :npv].copy(), F[npv:, npv:].copy(), ok

    def _backsolve(self, s, L, xe):
        npv, nb = int(self.npiv[s]), int(self.nbnd[s])
        f = npv + nb
        xb = xe[self.bnd_rows[self.bnd_off[s]:self.bnd_off[s] + nb]]
        w = L[f, :npv] - L[npv:f, :].T @ xb
        for c in range(npv - 1, -1, -1):
            w[c] = (w[c] - L[c + 1:npv, c] @ w[c + 1:npv]) / L[c, c]
        xe[self.piv0[s]:self.piv0[s] + npv] = w

    def _per_vertex(self, xe):
        dpose = np.zeros((self.n_poses, 3)); dlm = np.zeros((self.n_lms, 2))
        for p in range(self.n_poses):
            g = self.pose_gidx[p]
            if g >= 0:
                dpose[p] = xe[g:g + 3]
        for l in range(self.n_lms):
            g = self.lm_gidx[l]
            if g >= 0:
                dlm[l] = xe[g:g + 2]
        return dpose, dlm

    # ---- single-GPU replay
    def solve(self, blocks):
        """blocks: dict with Hpp_diag [N,9], Hll_diag [M,4], Hpp_off [Epp,9], Hpl [Epl,6], b_pose [N,3], b_lm [M,2]
        in INSERTION order (as the oracle's linearize_blocks returns them).  Returns (dpose [N,3], dlm [M,2], ok)."""
        Hpl, Hpp_off = self._device_blocks(blocks)
        S = self.n_fronts
        Ls, Us = [None] * S, [None] * S
        ok = True
        for s in range(S):                      # elimination order = children before parents
            F = self._assemble(s, blocks, Hpl, Hpp_off)
            self._extend_add(s, F, Us, lambda c: True)
            Ls[s], Us[s], good = self._factor(s, F)
            ok = ok and good
        xe = np.zeros(self.n_scalar)
        for s in range(S - 1, -1, -1):          # parents before children
            self._backsolve(s, Ls[s], xe)
        dpose, dlm = self._per_vertex(xe)
        return dpose, dlm, ok

    # ---- pose-window shards: what one rank does around the all-reduce
    def shard_local(self, blocks_of_rank):
        """blocks_of_rank: H blocks from ONLY the edges this rank evaluates (pl_rank/pp_rank == rank).
        Factorises the rank's own fronts and returns its contribution to the exchange buffer."""
        Hpl, Hpp_off = self._device_blocks(blocks_of_rank)
        S = self.n_fronts
        self._Ls, self._Us = [None] * S, [None] * S
        X = np.zeros(self.exchange_doubles)
        ok = True
        for s in range(S):
            if self.owner[s] == self.rank:
                F = self._assemble(s, blocks_of_rank, Hpl, Hpp_off)
                self._extend_add(s, F, self._Us, lambda c: True)
                self._Ls[s], self._Us[s], good = self._factor(s, F)
                ok = ok and good
        for s in range(S):
            if self.owner[s] < 0:
                F = self._assemble(s, blocks_of_rank, Hpl, Hpp_off)
                self._extend_add(s, F, self._Us, lambda c: self.owner[c] == self.rank)
                f = int(self.npiv[s] + self.nbnd[s])
                X[self.x_off[s]:self.x_off[s] + (f + 1) * f] = self._pack(F, f)
        return X, ok

    @staticmethod
    def _pack(F, f):
        """(f+1) x f column-major with ld = f+1, upper triangle zeroed (the exchange slot layout)."""
        G = F.copy()
        for c in range(f):
            G[:c, c] = 0.0
        return G.T.reshape(-1)                  # column c occupies [c*(f+1), (c+1)*(f+1))

    def shard_finish(self, X):
        """X: the all-reduced exchange buffer.  Returns (dpose, dlm, ok) valid for the vertices this rank knows."""
        S = self.n_fronts
        ok = True
        for s in range(S):
            if self.owner[s] < 0:
                f = int(self.npiv[s] + self.nbnd[s])
                F = X[self.x_off[s]:self.x_off[s] + (f + 1) * f].reshape(f, f + 1).T.copy()
                self._extend_add(s, F, self._Us, lambda c: self.owner[c] < 0)
                self._Ls[s], self._Us[s], good = self._factor(s, F)
                ok = ok and good
        xe = np.zeros(self.n_scalar)
        for s in range(S - 1, -1, -1):
            if self._Ls[s] is not None:
                self._backsolve(s, self._Ls[s], xe)
        dpose, dlm = self._per_vertex(xe)
        dpose[~self.pose_known] = 0.0; dlm[~self.lm_known] = 0.0
        return dpose, dlm, ok
