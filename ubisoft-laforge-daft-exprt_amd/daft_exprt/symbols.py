"""Symbol table of the English front-end.

Mirrors the table of the reference (`src/daft_exprt/symbols.py:18-36`): pad `_` at
index 0 (the collate zero-pads symbol ids, so index 0 must be the pad), then eos,
whitespace, the four punctuation marks and the stressed ARPAbet set -- 76 symbols.
Only the table itself is on the hot path (`hparams.n_symbols` sizes the embedding).
The ARPAbet inventory is generated rather than spelled out: 15 vowels x stress {0,1,2}
plus 24 consonants, alphabetically merged.
"""

pad = '_'
eos = '~'
whitespace = ' '
punctuation = ',.!?'

# silence / unknown markers used by the aligner files (kept for the data-loader row)
SIL_WORD_SYMBOL = '<sil>'
SIL_PHONE_SYMBOL = 'SIL'

_VOWELS = ('AA', 'AE', 'AH', 'AO', 'AW', 'AY', 'EH', 'ER', 'EY', 'IH', 'IY', 'OW', 'OY', 'UH', 'UW')
_CONSONANTS = ('B', 'CH', 'D', 'DH', 'F', 'G', 'HH', 'JH', 'K', 'L', 'M', 'N', 'NG', 'P', 'R',
               'S', 'SH', 'T', 'TH', 'V', 'W', 'Y', 'Z', 'ZH')


def _arpabet_stressed():
    phones = [(v, [f'{v}{s}' for s in range(3)]) for v in _VOWELS] + [(c, [c]) for c in _CONSONANTS]
    phones.sort(key=lambda kv: kv[0])
    return [p for _, group in phones for p in group]


arpabet_stressed = _arpabet_stressed()
symbols_english = list(pad + eos + whitespace + punctuation) + arpabet_stressed
assert len(symbols_english) == 76 and symbols_english[0] == pad
