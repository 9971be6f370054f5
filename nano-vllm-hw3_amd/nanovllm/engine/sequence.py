"""The per-request record of the engine: what the scheduler, the KV block allocator and the model runner need to
know about one prompt and its continuation (the counterpart of the reference's nanovllm/engine/sequence.py; no
prefix-cache fields and no pickling hooks, because every rank keeps its own copy instead of receiving it by RPC)."""
import itertools

from nanovllm.sampling_params import SamplingParams

_ids = itertools.count()

# life cycle of a request
QUEUED, ACTIVE, DONE = "queued", "active", "done"


class Sequence:
    __slots__ = ("seq_id", "state", "token_ids", "num_tokens", "num_prompt_tokens", "last_token", "block_table",
                 "max_tokens", "ignore_eos")

    def __init__(self, token_ids, sampling_params: SamplingParams = None):
        sp = sampling_params if sampling_params is not None else SamplingParams()
        ids = list(token_ids)
        assert ids, "a request needs at least one prompt token"
        self.seq_id = next(_ids)
        self.state = QUEUED
        self.token_ids = ids                    # prompt followed by the generated tokens
        self.num_prompt_tokens = self.num_tokens = len(ids)
        self.last_token = ids[-1]
        self.block_table = []                   # KV blocks, in order of the tokens they hold
        self.max_tokens = sp.max_tokens         # (the sampler is greedy: the temperature is not used)
        self.ignore_eos = sp.ignore_eos

    def __len__(self):
        return self.num_tokens

    def append_token(self, token_id: int):
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    @property
    def is_finished(self):
        return self.state == DONE

    @property
    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    @property
    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    def tokens_in_block(self, i: int, block_size: int) -> int:
        """How many of this sequence's tokens live in its i-th KV block."""
        return min(block_size, self.num_tokens - i * block_size)
