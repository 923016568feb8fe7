"""Several B=1 forwards in flight on one GPU.

The reference fills a GPU by starting two Ray workers per device, each with its own pickled copy
of the model (`ray.remote(num_gpus=0.5)`, src/inference/inference_OnePosePlus_worker.py:70,
src/inference/inference_OnePosePlus.py:62-99).  The MI355X-native equivalent needs no extra
processes: `MatcherPool` keeps `n_streams` forwards in flight on separate HIP streams of ONE
process (one module instance = one workspace per stream, weights loaded from the same state
dict), driven by one host thread per stream (the forward's single D2H sync of the match count
releases the GIL).  Measured on MI355X at 512x512 x 5k points in the default bf16x3 arithmetic (round 3/4, DESIGN.md section 5,
`profiles/r03_ab_streams.txt`): 416 images/s with one forward in flight, 476 / 489 / 493 with 2 / 3 / 4 streams on the latency
tiles and 502-520 with 3 streams on the throughput tiles this pool selects (`bench.py` `throughput_tiles_leg`).
"""
import queue
import threading

import torch

from .model import OnePosePlus_model


class MatcherPool:
    def __init__(self, config, state_dict, device=None, n_streams=3, gemm_precision=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.models, self.streams = [], []
        for _ in range(max(1, int(n_streams))):
            m = OnePosePlus_model(config).eval()
            if gemm_precision is not None:
                m.set_gemm_precision(gemm_precision)
            if int(n_streams) > 1:
                m.set_tile_policy("throughput")      # several forwards share the chip: tiles chosen for least CU time
                m.set_fpn_overlap(False)             # ... and they are each other's overlap: no side streams inside a forward
            m.load_state_dict(state_dict, strict=True)
            self.models.append(m.to(self.device))
            self.streams.append(torch.cuda.Stream(device=self.device))

    def map(self, items, post=None):
        """Runs `model(data)` for every `data` dict of `items` (tensors already on the device) and
        returns the mutated dicts in input order.  `post(data)` (optional) runs on the worker thread
        right after the forward, on the forward's stream (e.g. pose.ransac_PnP).

        Stream contract: the inputs must have been produced on (or be complete with respect to) the stream that
        is current when `map` is called -- e.g. `ingest.read_grayscale_u8` uploads and resizes asynchronously on
        it.  Every worker stream first waits for that stream, so a forward never reads a half-written image or
        bank; on return all worker streams are synchronised."""
        items = list(items)
        producer = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(producer)
        out = [None] * len(items)
        todo = queue.SimpleQueue()
        for i, d in enumerate(items):
            todo.put((i, d))
        errors = []

        def worker(slot):
            torch.cuda.set_device(self.device)
            try:
                self.streams[slot].wait_event(ready)      # inputs enqueued on the caller's stream are complete first
                with torch.cuda.stream(self.streams[slot]), torch.no_grad():
                    while True:
                        try:
                            i, d = todo.get_nowait()
                        except queue.Empty:
                            break
                        self.models[slot](d)
                        if post is not None:
                            post(d)
                        out[i] = d
                self.streams[slot].synchronize()
            except Exception as e:   # surfaced to the caller: nothing is swallowed
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(self.models))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out
