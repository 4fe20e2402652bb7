"""GPU reciprocal-rank fusion of retrieval lanes (hybrid /retrieve, BASELINE configs[4]) — the
device counterpart of _rrf_merge (/root/reference/app/retrieve.py:245-260, host mirror:
cadence_rag_amd.retrieve._rrf_merge)."""
from __future__ import annotations

import ctypes
from typing import Dict, Sequence, Tuple

import torch

from . import _native

DEFAULT_RRF_K = 60


_ext_streams: dict = {}


def _on_stream(stream: int, device):
    """Context in which torch allocates, uploads and frees on the caller's HIP stream, so that every
    temporary of a call is ordered with the kernels the C ABI enqueues on that same stream (stream 0 =
    the null stream, which is also torch's default current stream)."""
    key = (int(stream), str(device))
    ext = _ext_streams.get(key)
    if ext is None:   # (wrapping a raw stream handle costs ~10 us: once per stream)
        ext = _ext_streams[key] = (torch.cuda.ExternalStream(int(stream), device=device) if stream else
                                   torch.cuda.default_stream(device))
    return torch.cuda.stream(ext)


def rrf_fuse(lanes: Sequence[Tuple[torch.Tensor, torch.Tensor]], out_k: int, rrf_k: int = DEFAULT_RRF_K,
             stream: int = 0, out: "Dict[str, torch.Tensor] | None" = None) -> Dict[str, torch.Tensor]:
    """lanes: [(ids int64 [nq, width] CUDA, counts int32 [nq] CUDA), ...] in lane order (the
    reference's order is bm25, tech_tokens, dense).  Returns ids [nq, out_k] (-1 pad), scores fp64,
    lane-hit masks (bit l = lane l), counts.  `out`: a dict of that shape to write into (a caller that fuses the
    same batch shape again and again saves four allocations per call)."""
    lib = _native.load()
    n = len(lanes)
    nq = int(lanes[0][0].shape[0])
    dev = lanes[0][0].device
    with _on_stream(stream, dev):
        keep = [(t.contiguous(), c.contiguous()) for t, c in lanes]
        ids_arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(t.data_ptr()) for t, _ in keep])
        cnt_arr = (ctypes.c_void_p * n)(*[ctypes.c_void_p(c.data_ptr()) for _, c in keep])
        width = (ctypes.c_int * n)(*[int(t.shape[1]) for t, _ in keep])
        if out is None:
            out = {
                "ids": torch.empty(nq, out_k, dtype=torch.int64, device=dev),
                "scores": torch.empty(nq, out_k, dtype=torch.float64, device=dev),
                "lanes": torch.empty(nq, out_k, dtype=torch.int32, device=dev),
                "counts": torch.empty(nq, dtype=torch.int32, device=dev),
            }
        _native.check(lib.crag_rrf_fuse(n, ids_arr, cnt_arr, width, nq, int(rrf_k), int(out_k),
                                        out["ids"].data_ptr(), out["scores"].data_ptr(), out["lanes"].data_ptr(),
                                        out["counts"].data_ptr(), ctypes.c_void_p(stream)), "crag_rrf_fuse")
    return out


# ------------------------------------------------------------------------------------------------
# exact-token lane
# ------------------------------------------------------------------------------------------------
import hashlib  # noqa: E402
from array import array  # noqa: E402
from itertools import chain  # noqa: E402

import numpy as np  # noqa: E402

MAX_QUERY_TOKENS = 32


def token_hash(token: str) -> int:
    """64-bit hash of the EXACT token string (the SQL `&&` compares text[] elements exactly)."""
    return int.from_bytes(hashlib.blake2b(token.encode("utf-8"), digest_size=8).digest(), "little")


_ONE_ZERO = array("Q", [0])
_HASH_CACHE_MAX = 1 << 18
_hash_cache: dict = {}   # token string -> hash.  Query tokens repeat (ticket ids, error names, versions): a lookup is
                         # ~30 ns, blake2b + int.from_bytes ~1 us.  Dropped wholesale when it reaches the cap.


def _hashes(token_lists) -> "array":
    """All tokens of all lists, hashed, flat, in list order (array('Q'))."""
    cache = _hash_cache
    try:   # (flattening and looking up without a Python-level loop: half the time of the comprehension)
        return array("Q", list(map(cache.__getitem__, chain.from_iterable(token_lists))))
    except KeyError:
        if len(cache) > _HASH_CACHE_MAX:
            cache.clear()
        for toks in token_lists:
            for t in toks:
                if t not in cache:
                    cache[t] = token_hash(t)
        return array("Q", [cache[t] for toks in token_lists for t in toks])


class TechTokenIndex:
    """GPU-resident `tech_tokens text[]` column + the static order (call_started_at DESC, id ASC).
    Counterpart of _fetch_chunks_tech / _fetch_artifacts_tech (/root/reference/app/retrieve.py:183-242)."""

    def __init__(self, row_tokens, ids, call_started_at, device, verify: bool = True) -> None:
        """verify: keep the rows' token strings on the host and check every row the kernel returns against the
        query's strings.  The kernel compares 64-bit hashes; a collision (2^-64 per comparison) would otherwise be
        the one way this lane could differ from the SQL `&&`, which compares the strings themselves."""
        n = len(row_tokens)
        ids = np.asarray(ids, dtype=np.int64)
        ts = np.asarray(call_started_at, dtype="datetime64[us]").astype(np.int64)
        order = np.lexsort((ids, -ts)).astype(np.int32)  # primary: started_at DESC, then id ASC
        # CSR stored by rank (recency order) so the GPU scan is a coalesced stream
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(row_tokens[r]) for r in order], out=row_ptr[1:])
        toks = np.array([token_hash(t) for r in order for t in row_tokens[r]], dtype=np.uint64)
        if toks.size == 0:
            toks = np.zeros(1, dtype=np.uint64)
        self.n = n
        self.device = device
        self.order = torch.from_numpy(order).to(device)
        self.row_ptr = torch.from_numpy(row_ptr).to(device)
        self.tokens = torch.from_numpy(toks.view(np.int64)).to(device)
        self.ids = torch.from_numpy(ids).to(device)
        self._bitmaps: dict = {}   # per stream: two streams sharing one index must not share scratch
        self._pinned: dict = {}    # per stream: ring of pinned upload buffers for the query tokens
        self._rank_of_id = None
        self._order_host, self._ids_host = order, ids
        self._row_tokens = [frozenset(t) for t in row_tokens] if verify else None
        self._pos_of_id = None

    def _slot(self, stream: int):
        """Next upload slot of the stream's ring of four: a crag_upload_slot of the library (pinned host buffer for the
        query-token hashes [64, 32] + counts [64], its device twin, the event of its last upload) and the per-k output
        tensors.  Everything is allocated once per stream: a call allocates nothing and uploads with ONE copy.
        [Round 3: four torch.empty on the device per call, two copies, blake2b per token: 156 us of host time around
        22 us of kernels; round 4 first: a pinned torch buffer packed with numpy, 65 us; the packing, the copy and the
        event now live behind crag_tech_lane_host.]"""
        ring = self._pinned.setdefault(stream, {"next": 0, "slots": []})
        if len(ring["slots"]) < 4:
            with torch.cuda.device(self.device):
                handle = _native.load().crag_upload_slot_create()
            if not handle:
                raise _native.NativeLibraryError(f"crag_upload_slot_create failed: {_native.last_error()}")
            ring["slots"].append({"handle": handle, "out": {}})
        slot = ring["slots"][ring["next"] % len(ring["slots"])]
        ring["next"] += 1
        return slot

    def close(self) -> None:
        """Free the upload slots (pinned host memory).  Called by __del__ as well."""
        rings, self._pinned = getattr(self, "_pinned", {}), {}
        for ring in rings.values():
            for slot in ring["slots"]:
                _native.load().crag_upload_slot_destroy(slot["handle"])

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the library may be gone already
            pass

    _E2BIG = -4   # crag_dense.h CRAG_E2BIG: a query with more than 32 distinct tokens, nothing was enqueued

    def _pass(self, token_lists, k: int, row_mask, mask_stride: int, stream: int, borrow: bool = True):
        """One launch of the lane (crag_tech_lane_host: the hashes are packed, uploaded and matched behind ONE call).
        borrow: the returned tensors belong to the stream's upload ring and are valid until the fourth following call
        on the same stream; otherwise they are fresh.  Returns None when a query holds more than MAX_QUERY_TOKENS
        DISTINCT tokens (nothing was enqueued: the caller splits it into passes)."""
        nq = len(token_lists)
        if nq > 64:
            raise ValueError("the exact-token lane takes at most 64 queries per call")
        flat = _hashes(token_lists)
        lens = array("i", map(len, token_lists))
        if not len(flat):
            flat = _ONE_ZERO                     # (a valid address; no query reads it: all counts are 0)
        words = max((self.n + 63) // 64, 1)
        slot = self._slot(stream)
        bitmap = self._bitmaps.get(stream)
        if bitmap is None or bitmap.numel() < nq * words:
            with _on_stream(stream, self.device):
                bitmap = self._bitmaps[stream] = torch.empty(64 * words, dtype=torch.int64, device=self.device)
        if borrow:
            outs = slot["out"].get(k)
            if outs is None:
                with _on_stream(stream, self.device):
                    outs = slot["out"][k] = (torch.empty(64, k, dtype=torch.int64, device=self.device),
                                             torch.empty(64, dtype=torch.int32, device=self.device))
            out_ids, out_ct = outs[0][:nq], outs[1][:nq]
        else:   # fresh outputs: ONE allocation, the counts behind the ids -- on the caller's stream (entering a stream
                # context costs ~5 us: skipped when torch's current stream already is that stream)
            if torch.cuda.current_stream(self.device).cuda_stream == stream:
                block = torch.empty(nq * k + (nq + 1) // 2, dtype=torch.int64, device=self.device)
            else:
                with _on_stream(stream, self.device):
                    block = torch.empty(nq * k + (nq + 1) // 2, dtype=torch.int64, device=self.device)
            out_ids, out_ct = block[:nq * k].view(nq, k), block[nq * k:].view(torch.int32)[:nq]
        ptrs = self.__dict__.get("_ptrs")
        if ptrs is None:
            ptrs = self._ptrs = (self.order.data_ptr(), self.row_ptr.data_ptr(), self.tokens.data_ptr(), self.ids.data_ptr())
        rc = _native.load().crag_tech_lane_host(ptrs[0], ptrs[1], ptrs[2], ptrs[3], self.n, flat.buffer_info()[0],
                                                lens.buffer_info()[0], nq, int(k),
                                                None if row_mask is None else row_mask.data_ptr(), int(mask_stride),
                                                slot["handle"], bitmap.data_ptr(), out_ids.data_ptr(), out_ct.data_ptr(),
                                                ctypes.c_void_p(stream))
        if rc == self._E2BIG:
            return None
        _native.check(rc, "crag_tech_lane_host")
        return out_ids, out_ct

    def search(self, query_token_lists, k: int, row_mask=None, mask_stride: int = 0, stream: int = 0,
               verify: "bool | None" = None, borrow: bool = False):
        """verify (default: whatever the index was built with): check the returned rows' token STRINGS on the host.
        That check copies ids and counts to the host and therefore SYNCHRONISES the caller's stream; pass
        verify=False on a batched, stream-ordered path (HybridSearcher does) — the lane then differs from the SQL
        `&&` only by a 64-bit hash collision (2^-64 per comparison).
        query_token_lists: per query the tokens of extract_tech_tokens(query) (at most 64 queries), any
        number of tokens per query: the SQL `tech_tokens && :tokens` has no bound either.  The kernel takes 32
        tokens per query and launch; a longer list (a pasted log with many URLs / hashes) runs in several
        passes whose hits are merged in the lane's static order.
        borrow=True returns the lane's own ring buffers (valid until the fourth following call on the same stream:
        what a caller that consumes them at once wants -- HybridSearcher feeds them to the fusion kernel); the
        default returns copies.
        Returns (ids int64 [nq, k] -1 padded, counts int32 [nq]) CUDA tensors, best (most recent) first."""
        check = (self._row_tokens is not None) if verify is None else (bool(verify) and self._row_tokens is not None)
        if not check:
            # the common, stream-ordered path: the library drops repeated tokens itself and says so when a query
            # needs more than one pass
            fast = self._pass(query_token_lists, k, row_mask, mask_stride, stream, borrow)
            if fast is not None:
                return fast
        lists = [list(dict.fromkeys(toks)) for toks in query_token_lists]  # distinct, first occurrence kept
        passes = max(1, max((-(-len(t) // MAX_QUERY_TOKENS) for t in lists), default=1))
        if passes == 1:
            out_ids, out_ct = self._pass(lists, k, row_mask, mask_stride, stream, borrow)
            return self._verified(lists, k, out_ids, out_ct, row_mask, mask_stride, stream) if check else (out_ids, out_ct)
        if self._rank_of_id is None:
            order = self.order.cpu().numpy()
            ids = self.ids.cpu().numpy()
            self._rank_of_id = {int(ids[pos]): r for r, pos in enumerate(order)}
        hits = [set() for _ in lists]
        for p in range(passes):
            part = [t[p * MAX_QUERY_TOKENS:(p + 1) * MAX_QUERY_TOKENS] for t in lists]
            ids_p, ct_p = self._pass(part, k, row_mask, mask_stride, stream)
            with _on_stream(stream, self.device):
                ids_h, ct_h = ids_p.cpu().numpy(), ct_p.cpu().numpy()
            for q in range(len(lists)):
                hits[q].update(int(v) for v in ids_h[q, :ct_h[q]])
        out_ids = np.full((len(lists), k), -1, dtype=np.int64)
        out_ct = np.zeros(len(lists), dtype=np.int32)
        for q, found in enumerate(hits):
            best = sorted(found, key=self._rank_of_id.__getitem__)[:k]  # each pass returned ITS first k
            out_ids[q, :len(best)] = best
            out_ct[q] = len(best)
        with _on_stream(stream, self.device):
            d_ids, d_ct = torch.from_numpy(out_ids).to(self.device), torch.from_numpy(out_ct).to(self.device)
        if check:
            d_ids, d_ct = self._verified(lists, k, d_ids, d_ct, row_mask, mask_stride, stream)
        return d_ids, d_ct

    def _verified(self, lists, k, out_ids, out_ct, row_mask, mask_stride, stream):
        """String check of the hash matches (one small D2H copy per call).  A false positive — never observed,
        2^-64 per comparison — is repaired by evaluating that query on the host from the strings."""
        with _on_stream(stream, self.device):
            ids_h, ct_h = out_ids.cpu().numpy(), out_ct.cpu().numpy()
        if self._pos_of_id is None:
            self._pos_of_id = {int(v): i for i, v in enumerate(self._ids_host)}
        bad = [q for q, toks in enumerate(lists)
               if any(self._row_tokens[self._pos_of_id[int(r)]].isdisjoint(toks) for r in ids_h[q, :ct_h[q]])]
        if not bad:
            return out_ids, out_ct
        mask_h = None
        if row_mask is not None:
            with _on_stream(stream, self.device):
                mask_h = row_mask.cpu().numpy().reshape(-1)
        for q in bad:
            toks, hits = set(lists[q]), []
            for pos in self._order_host:
                if mask_h is not None:
                    byte = mask_h[(q * mask_stride if mask_stride else 0) + (int(pos) >> 3)]
                    if not (byte >> (int(pos) & 7)) & 1:
                        continue
                if not self._row_tokens[pos].isdisjoint(toks):
                    hits.append(int(self._ids_host[pos]))
                    if len(hits) == k:
                        break
            ids_h[q] = -1
            ids_h[q, :len(hits)] = hits
            ct_h[q] = len(hits)
        with _on_stream(stream, self.device):
            return torch.from_numpy(ids_h).to(self.device), torch.from_numpy(ct_h).to(self.device)


# ------------------------------------------------------------------------------------------------
# batched hybrid retrieve (BASELINE configs[4]): dense + exact-token + given BM25 lanes -> RRF, on the GPU
# ------------------------------------------------------------------------------------------------
class HybridSearcher:
    """The candidate stage of retrieve_evidence (/root/reference/app/retrieve.py:437-545) for a BATCH of up
    to 64 queries over one table, with every lane and the fusion on the device and no host round trip in
    between: dense top-`dense_k` (exact cosine scan), exact-token top-`tech_k`, caller-supplied BM25 ids
    (pg_search's ranking is an input, as it is to _rrf_merge), fused by reciprocal rank in the reference's
    lane order bm25 -> tech_tokens -> dense."""

    def __init__(self, index, tech_index: "TechTokenIndex | None" = None, *, dense_k: int = 50, tech_k: int = 50,
                 rrf_k: int = DEFAULT_RRF_K, verify_tokens: bool = False, overlap_lanes: bool = True) -> None:
        """verify_tokens: run the exact-token lane's host-side string check (a blocking D2H copy per step); off by
        default so that a step only enqueues work on the caller's stream.
        overlap_lanes: run the exact-token lane on a side stream beside the dense search (forked from the caller's
        stream behind the scan's launch, joined in front of the fusion kernel; same results, tested).  ON by default
        since round 4: measured on MI355X in alternating rounds (1M chunks, 64 queries, dense top-100) 0.395 ms per
        step against 0.421 with the lanes in series.  [Round 3 measured 0.440 vs 0.441 and left it off: the scan holds
        every CU (one 512-thread workgroup each), so the token lane's workgroups start only as scan workgroups retire;
        what they overlap with now is the selection launch, which occupies a quarter of the chip for ~22 us.]"""
        self.index, self.tech = index, tech_index
        self.verify_tokens = bool(verify_tokens)
        self.overlap_lanes = bool(overlap_lanes)
        self.dense_k, self.tech_k, self.rrf_k = int(dense_k), int(tech_k), int(rrf_k)
        self._dense_out: dict = {}  # per (stream, batch size): results of calls on different streams stay apart
        self._fused_out: dict = {}  # per (stream, batch size, out_k): the fusion's outputs
        self._side: dict = {}       # per caller stream: (side stream, fork event, join event)

    def search(self, query_vectors: torch.Tensor, query_token_lists=None, bm25=None, *, out_k: int = 0,
               row_mask=None, mask_stride: int = 0, stream: int = 0) -> Dict[str, torch.Tensor]:
        """query_vectors [nq, dim] fp32 CUDA; query_token_lists: per query its extract_tech_tokens();
        bm25: (ids int64 [nq, w] CUDA, counts int32 [nq] CUDA) or None; row_mask: packed bits per row
        position (uint8 CUDA), shared (mask_stride 0) or per query.  Returns rrf_fuse's dict plus the dense
        lane itself ("dense_ids", "dense_scores", "dense_counts").  Everything is enqueued on `stream` (token
        lists longer than 32 tokens and verify_tokens=True are the exceptions: both visit the host); the
        dense and fused buffers are reused by the next call with the same stream and batch shape (the returned
        tensors are valid until then), and the caller orders `query_vectors` / `row_mask` / `bm25` with `stream`."""
        nq = int(query_vectors.shape[0])
        dev = query_vectors.device
        if (stream, nq) not in self._dense_out:
            with _on_stream(stream, dev):
                self._dense_out[(stream, nq)] = (torch.empty(nq, self.dense_k, dtype=torch.int64, device=dev),
                                                 torch.empty(nq, self.dense_k, dtype=torch.float32, device=dev),
                                                 torch.empty(nq, dtype=torch.int32, device=dev))
        d_ids, d_sc, d_ct = self._dense_out[(stream, nq)]
        use_tech = self.tech is not None and query_token_lists is not None
        tech_lane = None
        overlap = use_tech and self.overlap_lanes and not self.verify_tokens
        if overlap:
            # fork: everything the caller enqueued so far (queries, masks, the previous step's fusion, which may
            # still read the side stream's recycled buffers) is ordered before the side stream's work
            if stream not in self._side:
                self._side[stream] = (torch.cuda.Stream(device=dev), torch.cuda.Event(), torch.cuda.Event())
            side, fork, join = self._side[stream]
            with _on_stream(stream, dev):
                fork.record()
        self.index.search_async(query_vectors, self.dense_k, d_ids, d_sc, d_ct, d_row_mask=row_mask,
                                mask_stride=mask_stride, stream=stream)
        if overlap:  # the dense scan is enqueued first: the host's share of the token lane runs under it
            side.wait_event(fork)
            tech_lane = self.tech.search(query_token_lists, self.tech_k, row_mask=row_mask, mask_stride=mask_stride,
                                         stream=side.cuda_stream, verify=False, borrow=True)
            join.record(side)
        lanes = []
        if bm25 is not None:
            lanes.append(bm25)
        if tech_lane is not None:
            with _on_stream(stream, dev):
                torch.cuda.current_stream(dev).wait_event(self._side[stream][2])   # join in front of the fusion
            lanes.append(tech_lane)
        elif use_tech:
            lanes.append(self.tech.search(query_token_lists, self.tech_k, row_mask=row_mask, mask_stride=mask_stride,
                                          stream=stream, verify=self.verify_tokens, borrow=True))
        lanes.append((d_ids, d_ct))
        width = sum(int(t.shape[1]) for t, _ in lanes)
        ok = out_k or width
        fused = self._fused_out.get((stream, nq, ok))
        if fused is None:   # like the dense buffers: reused by the next call with the same stream and batch shape
            with _on_stream(stream, dev):
                fused = self._fused_out[(stream, nq, ok)] = {
                    "ids": torch.empty(nq, ok, dtype=torch.int64, device=dev),
                    "scores": torch.empty(nq, ok, dtype=torch.float64, device=dev),
                    "lanes": torch.empty(nq, ok, dtype=torch.int32, device=dev),
                    "counts": torch.empty(nq, dtype=torch.int32, device=dev)}
        out = dict(rrf_fuse(lanes, out_k=ok, rrf_k=self.rrf_k, stream=stream, out=fused))
        out["dense_ids"], out["dense_scores"], out["dense_counts"] = d_ids, d_sc, d_ct
        return out
