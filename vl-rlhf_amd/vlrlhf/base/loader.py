"""Input pipeline of the DPO step.  In the reference the collator - JPEG decode + CLIP preprocess of
B images (src/vlrlhf/models/Llava/__init__.py:435-443) - runs on the training thread inside the step (HF dataloader with
dataloader_num_workers = 0).  Here a background thread collates `depth` batches ahead and pins them; the training thread
only issues the host-to-device copy, on a dedicated copy stream, while the GPU is still busy with the previous step."""
import queue
import threading
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional

import torch

_END = object()


def _map_tensors(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    return obj


class PrefetchLoader:
    def __init__(self, row_batches: Callable[[], Iterable[List[dict]]], collate: Callable[[List[dict]], Dict[str, Any]],
                 device: Optional[torch.device] = None, depth: int = 2):
        self.row_batches, self.collate = row_batches, collate
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda" and torch.cuda.is_available()
        self.depth = max(1, int(depth))
        self._copy_stream = torch.cuda.Stream(self.device) if self.cuda else None

    def _worker(self, q: "queue.Queue", stop: threading.Event):
        try:
            for rows in self.row_batches():
                if stop.is_set():
                    return
                batch = self.collate(rows)
                if self.cuda:
                    batch = _map_tensors(batch, lambda t: t.pin_memory() if not t.is_pinned() else t)
                while not stop.is_set():
                    try:
                        q.put(batch, timeout=0.1)
                        break
                    except queue.Full:
                        continue
            q.put(_END)
        except BaseException as e:      # surfaced on the training thread
            q.put(e)

    def _to_device(self, batch):
        if not self.cuda:
            return batch
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            out = _map_tensors(batch, lambda t: t.to(self.device, non_blocking=True))
        main.wait_stream(self._copy_stream)
        _map_tensors(out, lambda t: t.record_stream(main) or t)
        # tensor attributes the collator may have set (e.g. the image-duplication tag) do not survive .to(): none are set
        # before concatenated_inputs, which runs on the device copy
        return out

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._worker, args=(q, stop), name="vlr-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self._to_device(item)
        finally:
            stop.set()
            while th.is_alive():            # unblock a producer stuck on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
