"""Token table of the reference front-end (text_utils.py:3-26): phonemised text (IPA string from phonemizer/espeak,
host side, outside the engine) -> token ids for `TextEncoder` / PL-BERT.  The symbol inventory IS the interface to the
published checkpoints (n_token = 178), so it is restated verbatim; everything else is the engine's own code."""

_PAD = "$"
_PUNCTUATION = ';:,.!?¡¿—…"«»“” '
_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_LETTERS_IPA = ("ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁǂǃˈˌːˑʼʴʰʱʲʷˠˤ˞"
                "↓↑→↗↘'̩'ᵻ")

SYMBOLS = [_PAD] + list(_PUNCTUATION) + list(_LETTERS) + list(_LETTERS_IPA)
SYMBOL_TO_ID = {}
for _i, _s in enumerate(SYMBOLS):
    SYMBOL_TO_ID[_s] = _i  # later duplicates win, as in the reference's dict construction (text_utils.py:13-15)


class TextCleaner:
    """`TextCleaner()(phonemes) -> [ids]`; characters outside the table are dropped (the reference prints the text and
    skips them, text_utils.py:20-25).  `encode` adds the leading pad id the notebooks insert (ipynb:277)."""

    def __init__(self, dummy=None):
        self.word_index_dictionary = SYMBOL_TO_ID

    def __call__(self, text):
        return [self.word_index_dictionary[c] for c in text if c in self.word_index_dictionary]

    def encode(self, text):
        return [0] + self(text)
