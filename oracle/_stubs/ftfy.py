def fix_text(t):
    return t
