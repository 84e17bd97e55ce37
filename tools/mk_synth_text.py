"""synthetic lyric-like text for a training-dynamics sanity run (no dataset travels to the GPU box): python tools/mk_synth_text.py"""
import random
random.seed(0)
words = ["started", "from", "the", "bottom", "now", "we", "here", "hold", "on", "going", "home", "one", "dance", "hotline", "bling", "god", "plan", "passion", "fruit", "nice", "for", "what", "in", "my", "feelings", "you", "know", "it", "love", "money", "night", "city", "time", "never", "always", "back", "up", "down", "way", "too", "much", "good", "girl", "take", "care"]
lines = []
for _ in range(40000):
    n = random.randint(4, 9)
    w = [random.choice(words) for _ in range(n)]
    lines.append(" ".join(w).capitalize() + random.choice([",", ".", "", "!"]))
open("gpurun_out/synth_lyrics.txt", "w").write("\n".join(lines))
print(sum(len(l) + 1 for l in lines))
