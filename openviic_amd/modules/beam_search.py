"""Host-loop beam search over the step-wise model API (reference ``models/modules/beam_search.py``).

This is the compatibility driver: one ``model.step`` per time step, candidate selection and state
re-ordering on the host stream with native operators.  The production path is the fused engine
(``BaseTransformer.beam_search(fused=True)``); this class exists so that ``step`` /
``statefulness`` / ``apply_to_states`` keep the reference's observable behaviour.

Semantics (``beam_search.py:41-118``): one live beam at t=0; a beam that has emitted <eos> is
frozen -- its only viable continuation is word 0 with unchanged score, every other word scores
-999; candidates are ranked over the flattened (beam, word) axis; all per-beam state follows the
selected beams; after ``max_len`` steps beams are ordered by total score.
"""
import torch

from .. import ops


class BeamSearch:
    def __init__(self, model, b_s: int, max_len: int, eos_idx: int, beam_size: int, device):
        self.model, self.b_s, self.max_len = model, b_s, max_len
        self.eos_idx, self.beam_size, self.device = eos_idx, beam_size, device

    def _follow(self, beams, width):
        B, k = self.b_s, self.beam_size

        def fn(state):
            tail = list(state.shape[1:])
            index = beams.view(B, k, *([1] * len(tail))).expand(B, k, *tail)
            return torch.gather(state.view(B, width, *tail), 1, index).reshape(B * k, *tail)
        return fn

    def apply(self, out_size=1, return_probs=False, **kwargs):
        B, k, T = self.b_s, self.beam_size, self.max_len
        alive = torch.ones(B, k, 1, device=self.device)
        running = torch.zeros(B, 1, 1, device=self.device)
        words, history, step_logp, all_logp = None, [], [], []
        for t in range(T):
            width = 1 if t == 0 else k
            logp = self.model.step(t, words, **kwargs).view(B, width, -1)
            V = logp.shape[-1]
            prev = None if t == 0 else words.view(B, width)
            chosen, score, logp, alive = ops.beam_select(logp, running, alive, prev, self.eos_idx, k)
            beams = torch.div(chosen, V, rounding_mode="trunc")
            new_words = chosen - beams * V
            self.model.apply_to_states(self._follow(beams, width))
            running = score.unsqueeze(-1)
            alive = torch.gather(alive, 1, beams.unsqueeze(-1))
            history = [torch.gather(o, 1, beams.unsqueeze(-1)) for o in history] + [new_words.unsqueeze(-1)]
            if return_probs:
                all_logp.append((logp.expand(B, k, V) if t == 0 else logp).unsqueeze(2))
            picked = torch.gather(logp, 1, beams.unsqueeze(-1).expand(B, k, V))
            picked = torch.gather(picked, 2, new_words.unsqueeze(-1))
            step_logp = [torch.gather(o, 1, beams.unsqueeze(-1)) for o in step_logp] + [picked]
            words = new_words.reshape(-1, 1)
        running, order = torch.sort(running, dim=1, descending=True, stable=True)
        outputs = torch.gather(torch.cat(history, -1), 1, order.expand(B, k, T))[:, :out_size].contiguous()
        log_probs = torch.gather(torch.cat(step_logp, -1), 1, order.expand(B, k, T))[:, :out_size].contiguous()
        if out_size == 1:
            outputs, log_probs = outputs.squeeze(1), log_probs.squeeze(1)
        if return_probs:
            everything = torch.gather(torch.cat(all_logp, 2), 1,
                                      order.unsqueeze(-1).expand(B, k, T, all_logp[0].shape[-1]))
            return outputs, log_probs, everything
        return outputs, log_probs
