"""String helpers (mirror of xmca/tools/text.py)."""
import textwrap


def secure_str(string):
    return string.lower().replace(' ', '_')


def boldify_str(string):
    import matplotlib.pyplot as plt
    if plt.rcParams['text.usetex']:
        return ''.join([r'\textbf{', string, '}'])
    return string


def wrap_str(string):
    return textwrap.indent(textwrap.fill(string, width=80), '# ')
