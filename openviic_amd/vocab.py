"""Token ids -> caption strings (the step after the hot path).

``WordVocab`` carries exactly what the model and the decoding step read from the reference's
``Vocab`` (``data_utils/vocab.py:41-66,94-122,135-136``): the four special tokens in slots 0..3,
``itos`` / ``stoi``, ``max_caption_length`` and ``__len__``.  Building a vocabulary from dataset
annotations is out of scope; construct it from an existing word list (e.g. ``reference_vocab.itos``).
"""
import itertools
from typing import Iterable, List, Sequence

import torch


class WordVocab:
    def __init__(self, itos: Sequence[str], max_caption_length: int, padding_token="<pad>", bos_token="<bos>",
                 eos_token="<eos>", unk_token="<unk>"):
        self.itos = list(itos)
        self.stoi = {tok: i for i, tok in enumerate(self.itos)}
        self.padding_token, self.bos_token, self.eos_token, self.unk_token = padding_token, bos_token, eos_token, unk_token
        self.specials = [padding_token, bos_token, eos_token, unk_token]
        for tok in self.specials:
            if tok not in self.stoi:
                raise ValueError("special token {!r} is not in the word list".format(tok))
        self.padding_idx, self.bos_idx = self.stoi[padding_token], self.stoi[bos_token]
        self.eos_idx, self.unk_idx = self.stoi[eos_token], self.stoi[unk_token]
        self.max_caption_length = int(max_caption_length)

    def __len__(self) -> int:
        return len(self.itos)

    def encode_caption(self, caption: Iterable[str]) -> torch.Tensor:
        """``<bos> words <eos>`` padded to ``max_caption_length`` (``data_utils/vocab.py:96-102``)."""
        vec = torch.full((self.max_caption_length,), self.padding_idx, dtype=torch.long)
        for i, token in enumerate([self.bos_token] + list(caption) + [self.eos_token]):
            vec[i] = self.stoi.get(token, self.unk_idx)
        return vec

    def decode_caption(self, caption_vecs, join_words: bool = True):
        """Ids ``(bs, T)`` -> captions: special tokens are dropped and decoding stops at the first
        ``<eos>`` (``data_utils/vocab.py:104-122``)."""
        if isinstance(caption_vecs, torch.Tensor):
            caption_vecs = caption_vecs.detach().cpu().tolist()
        captions = []
        for vec in caption_vecs:
            words: List[str] = []
            for idx in vec:
                word = self.itos[idx]
                if word not in self.specials:
                    words.append(word)
                if idx == self.eos_idx:
                    break
            captions.append(" ".join(words) if join_words else " ".join(words).strip().split())
        return captions


def collapse_repeated_words(words: Sequence[str]) -> str:
    """The trainer's post-processing of a generated caption: consecutive duplicate words are merged
    (``itertools.groupby``, ``trainers/vi_trainer.py:251``)."""
    return " ".join(key for key, _ in itertools.groupby(words))


def captions_from_ids(vocab: WordVocab, ids) -> List[str]:
    """``beam_search`` output ``(B, T)`` -> final caption strings as the reference's prediction loop
    produces them (``trainers/vi_trainer.py:247-252``)."""
    if isinstance(ids, torch.Tensor):
        ids = ids.contiguous().view(-1, ids.shape[-1])
    return [collapse_repeated_words(words) for words in vocab.decode_caption(ids, join_words=False)]
