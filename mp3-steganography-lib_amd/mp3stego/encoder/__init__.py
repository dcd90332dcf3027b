from mp3stego.encoder.encoder import Encoder
