"""Drop-in import path of the reference's ``modules/midi_tokenizer.py``."""
from adt_str_amd.tokenizer import MidiTokenizer, MidiTokenizerConfig  # noqa: F401
