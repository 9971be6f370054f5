from collections import deque


class BlockManager:
    """Free-list allocator of fixed-size KV blocks.  The reference additionally de-duplicates full
    blocks by content hash (prefix cache, nanovllm/engine/block_manager.py:36-82); that is control-
    plane work outside the quantized-linear path and is not reproduced: every sequence owns its blocks."""

    def __init__(self, num_blocks: int, block_size: int):
        self.block_size = block_size
        self.num_blocks = num_blocks
        self.free = deque(range(num_blocks))

    def _needed(self, seq):
        return (len(seq) + self.block_size - 1) // self.block_size

    def can_allocate(self, seq) -> bool:
        return len(self.free) >= self._needed(seq)

    def allocate(self, seq):
        assert not seq.block_table
        for _ in range(self._needed(seq)):
            seq.block_table.append(self.free.popleft())

    def deallocate(self, seq):
        self.free.extend(reversed(seq.block_table))
        seq.block_table.clear()

    def can_append(self, seq) -> bool:
        # a new block is needed when the token about to be written starts a block
        return len(self.free) >= (1 if len(seq) % self.block_size == 1 else 0)

    def may_append(self, seq):
        if len(seq) % self.block_size == 1 and len(seq.block_table) < self._needed(seq):
            seq.block_table.append(self.free.popleft())
