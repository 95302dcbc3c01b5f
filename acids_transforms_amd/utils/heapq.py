"""Host-side binary min-heap with the reference's ordering contract (utils/heapq.py:9-59), re-exported at package level
like the reference does (utils/__init__.py:2).

Items are compared through `item[0]` with strict `<` only.  `heappop` does not sift the last element down from the root:
it walks the hole down to a leaf, promoting at every level the LEFT child only if it is strictly smaller than the right
one (so the right child wins ties), drops the last element into that leaf and lets it rise.  That choice of path is what
fixes the pop order among equal keys, and the offline PGHI kernels (csrc/pghi.hip) reproduce it on the device; this
module is the plain-Python form of the same rule for callers of the reference's helper (`DGT.perform_hgi` there pushes
`(-magnitude, (frame, bin))` tuples).  No device work happens here.
"""
from typing import Any, List, Tuple

__all__ = ["heappush", "heappop"]

HeapItemType = Tuple[Any, Any]
HeapType = List[HeapItemType]


def _rise(heap: HeapType, floor: int, pos: int) -> None:
    """Move heap[pos] towards `floor` while it is strictly smaller than its parent."""
    item = heap[pos]
    while pos > floor:
        up = (pos - 1) // 2
        if not (item[0] < heap[up][0]):
            break
        heap[pos] = heap[up]
        pos = up
    heap[pos] = item


def heappush(heap: HeapType, item: HeapItemType) -> None:
    heap.append(item)
    _rise(heap, 0, len(heap) - 1)


def heappop(heap: HeapType) -> HeapItemType:
    last = heap.pop()               # IndexError on an empty heap, like list.pop
    if not heap:
        return last
    top = heap[0]
    n = len(heap)
    hole = 0
    while True:
        child = 2 * hole + 1
        if child >= n:
            break
        if child + 1 < n and not (heap[child][0] < heap[child + 1][0]):
            child += 1              # ties go to the right child
        heap[hole] = heap[child]
        hole = child
    heap[hole] = last
    _rise(heap, 0, hole)
    return top
